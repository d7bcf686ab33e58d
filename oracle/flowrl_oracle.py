"""CPU oracle for the FlowRL action-sampling hot path (numpy restatement).

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it;
the product path (``flow-rl_amd/``) never does.

PARITY UNPINNED: the reference (typoverflow/flow-rl 0.2.0) ships no tests, golden
vectors or fixtures for this path, and its arithmetic lives in third-party
packages that are absent from this container and from /root/reference
(jax==0.5.3, flax==0.10.5; ``import jax`` -> ModuleNotFoundError, an ordinary
missing-module error).  This file restates the reference *text* line by line;
what pins it instead is (1) float64 vs float32 agreement, (2) an independent
torch-CPU restatement of the network in tests/test_oracle.py, (3) closed-form
known-answer tests and the hand-computed table values of SURVEY.md appendix A.6.

Every function cites the reference file:line it follows (paths relative to the
reference checkout).  All arithmetic is done in ``dtype`` (float32 mirrors the
reference's JAX default; float64 is the ground truth used for tolerances).

Third-party semantics restated from their published definitions:
  * flax.linen.Dense:  y = x @ kernel + bias, kernel stored (in, out) row-major
  * flax.linen.softplus = logaddexp(x, 0); flax.linen.gelu(approximate=True)
    = 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
  * jax.nn.silu = x sigmoid(x); jax.nn.elu = x>0 ? x : expm1(x)
  * flax submodule auto-naming inside @nn.compact: attribute modules keep the
    attribute name (noise_predictor / vel_predictor / time_embedding /
    cond_embedding), the inline time MLP is ``MLP_0``, Dense layers ``Dense_i``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------- #
# activations — flowrl/functional/activation.py:8-18 (+ flax gelu for FQL,
# flowrl/agent/offline/fql/fql.py:199)
# --------------------------------------------------------------------------- #


def softplus(x):
    # flax.linen.softplus == jnp.logaddexp(x, 0)
    return np.logaddexp(x, np.zeros((), dtype=x.dtype))


def mish(x):
    # activation.py:17-18  x * tanh(softplus(x))
    return x * np.tanh(softplus(x))


def silu(x):
    one = np.ones((), dtype=x.dtype)
    return x / (one + np.exp(-x))


def relu(x):
    return np.maximum(x, np.zeros((), dtype=x.dtype))


def elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, np.zeros((), dtype=x.dtype))))


def gelu_tanh(x):
    dt = x.dtype
    c = np.asarray(math.sqrt(2.0 / math.pi), dtype=dt)
    k = np.asarray(0.044715, dtype=dt)
    half = np.asarray(0.5, dtype=dt)
    one = np.ones((), dtype=dt)
    return half * x * (one + np.tanh(c * (x + k * (x * x * x))))


ACTIVATIONS = {
    "mish": mish,
    "silu": silu,
    "relu": relu,
    "elu": elu,
    "tanh": np.tanh,
    "gelu_tanh": gelu_tanh,
}


def get_activation(name: str):
    # activation.py:8-15
    return ACTIVATIONS[name]


# --------------------------------------------------------------------------- #
# noise schedules — flowrl/flow/ddpm.py:23-42 and :132-150
# --------------------------------------------------------------------------- #


def linear_beta_schedule(T: int, beta_min=1e-4, beta_max=2e-2, dtype=np.float32):
    # ddpm.py:23-24
    return np.linspace(beta_min, beta_max, T, dtype=dtype)


def cosine_beta_schedule(T: int, s=0.008, dtype=np.float32):
    # ddpm.py:27-33
    steps = T + 1
    t = np.linspace(0, T, steps, dtype=dtype) / np.asarray(T, dtype=dtype)
    sd = np.asarray(s, dtype=dtype)
    ac = np.cos((t + sd) / (1 + sd) * np.asarray(np.pi, dtype=dtype) * np.asarray(0.5, dtype=dtype)) ** 2
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    return np.clip(betas, 0, 0.999).astype(dtype)


def vp_beta_schedule(T: int, dtype=np.float32):
    # ddpm.py:36-42
    t = np.arange(1, T + 1)
    b_max, b_min = 10.0, 0.1
    expo = (-b_min / T - 0.5 * (b_max - b_min) * (2 * t - 1).astype(dtype) / np.asarray(T ** 2, dtype=dtype)).astype(dtype)
    alpha = np.exp(expo)
    return (1 - alpha).astype(dtype)


def ddpm_tables(steps: int, noise_schedule: str, approx_postvar: bool = False,
                noise_schedule_params: Optional[dict] = None, dtype=np.float32):
    """betas/alphas/alpha_hats/postvars, each (T+1,) — ddpm.py:132-150."""
    kw = dict(noise_schedule_params or {})
    if noise_schedule == "linear":
        betas = linear_beta_schedule(steps, dtype=dtype, **kw)
    elif noise_schedule == "cosine":
        betas = cosine_beta_schedule(steps, dtype=dtype, **kw)
    elif noise_schedule == "vp":
        betas = vp_beta_schedule(steps, dtype=dtype)
    else:
        raise NotImplementedError(f"Unsupported noise schedule: {noise_schedule}")
    betas = np.concatenate([np.zeros((1,), dtype=dtype), betas.astype(dtype)])
    alphas = (1 - betas).astype(dtype)
    alpha_hats = np.cumprod(alphas, dtype=dtype)
    if approx_postvar:
        postvars = betas.copy()
    else:
        pv = betas[1:] * (1 - alpha_hats[:-1]) / (1 - alpha_hats[1:])
        postvars = np.concatenate([np.zeros((1,), dtype=dtype), pv.astype(dtype)])
    return betas, alphas, alpha_hats, postvars


# continuous-time schedules — flowrl/flow/continuous_ddpm.py:19-40


def linear_noise_schedule(t, beta0=0.1, beta1=20.0):
    # continuous_ddpm.py:19-24
    dt = t.dtype
    log_alpha = -np.asarray((beta1 - beta0) / 4.0, dtype=dt) * (t ** 2) - np.asarray(beta0 / 2.0, dtype=dt) * t
    alpha = np.exp(log_alpha)
    sigma = np.sqrt(1 - alpha ** 2)
    return alpha, sigma


def cosine_noise_schedule(t, s=0.008):
    # continuous_ddpm.py:26-31
    dt = t.dtype
    half_pi = np.asarray(np.pi / 2.0, dtype=dt)
    sd = np.asarray(s, dtype=dt)
    num = np.cos(half_pi * (np.clip(t, 0.0, 0.9946).astype(dt) + sd) / (1 + sd))
    den = np.cos(half_pi * sd / (1 + sd))
    alpha = (num / den).astype(dt)
    sigma = np.sqrt(1 - alpha ** 2)
    return alpha, sigma


def quad_t_schedule(T: int, n=1.0, tmin=1e-3, tmax=1.0, dtype=np.float32):
    # continuous_ddpm.py:35-40
    lin = np.linspace(0, 1, T + 1, dtype=dtype)
    return (np.asarray(tmax - tmin, dtype=dtype) * (lin ** np.asarray(n, dtype=dtype)) + np.asarray(tmin, dtype=dtype)).astype(dtype)


def continuous_tables(steps: int, noise_schedule: str, t_schedule_n=1.0, epsilon=1e-3,
                      noise_schedule_params: Optional[dict] = None, dtype=np.float32):
    """ts / alpha_hats / alphas / betas built per call — continuous_ddpm.py:84-87,134-138."""
    kw = dict(noise_schedule_params or {})
    if noise_schedule == "cosine":
        tmin, tmax = epsilon, 0.9946
        fn = cosine_noise_schedule
    elif noise_schedule == "linear":
        tmin, tmax = epsilon, 1.0
        fn = linear_noise_schedule
    else:
        raise NotImplementedError(f"Unsupported noise schedule: {noise_schedule}")
    ts = quad_t_schedule(steps, n=t_schedule_n, tmin=tmin, tmax=tmax, dtype=dtype)
    alpha_hats = (fn(ts, **kw)[0] ** 2).astype(dtype)
    alphas = np.concatenate([np.ones((1,), dtype=dtype), alpha_hats[1:] / alpha_hats[:-1]]).astype(dtype)
    betas = (1 - alphas).astype(dtype)
    return ts, alpha_hats, alphas, betas


# --------------------------------------------------------------------------- #
# time embeddings — flowrl/module/time_embedding.py:6-27
# --------------------------------------------------------------------------- #


def positional_embedding(x, output_dim: int, max_positions: int = 10000):
    # time_embedding.py:11-16; x has shape (..., 1)
    dt = x.dtype
    half = output_dim // 2
    f = np.asarray(np.log(np.asarray(max_positions, dtype=dt)) / (half - 1), dtype=dt)
    f = np.exp(np.arange(half).astype(dt) * -f).astype(dt)
    f = x * f
    return np.concatenate([np.cos(f), np.sin(f)], axis=-1).astype(dt)


def learnable_fourier_embedding(x, kernel):
    # time_embedding.py:23-27; kernel (D/2, in_dim)
    dt = x.dtype
    f = np.asarray(2 * np.pi, dtype=dt) * x @ kernel.T.astype(dt)
    return np.concatenate([np.cos(f), np.sin(f)], axis=-1).astype(dt)


# --------------------------------------------------------------------------- #
# networks — flowrl/module/mlp.py:10-34, flowrl/agent/offline/fql/network.py:10-35,
# flowrl/flow/ddpm.py:70-95, flowrl/flow/cnf.py:17-42
# --------------------------------------------------------------------------- #


def dense(p, x):
    return x @ p["kernel"] + p["bias"]


def layer_norm(p, x, eps=1e-6):
    """flax.linen.LayerNorm (flax 0.10.5 defaults: epsilon 1e-6, use_fast_variance: var = max(E[x^2] - E[x]^2, 0),
    y = (x - mean) * (rsqrt(var + eps) * scale) + bias), over the last axis."""
    mean = x.mean(axis=-1, keepdims=True)
    var = np.maximum((x * x).mean(axis=-1, keepdims=True) - mean * mean, 0)
    mul = (1 / np.sqrt(var + x.dtype.type(eps))) * p["scale"]
    return (x - mean) * mul + p["bias"]


def mlp_apply(p: Dict, x, n_hidden: int, activation, has_output: bool, ln: bool = False, ln_after_act: bool = False):
    """mlp.py:25-33, dropout off: Dense -> [LayerNorm] -> act per hidden layer (layer_norm is off in all shipped
    sampler configs; flax names the i-th LayerNorm of the module "LayerNorm_i").  ln_after_act: FQL's MLP
    (agent/offline/fql/network.py:26-30) puts the LayerNorm AFTER the activation."""
    for i in range(n_hidden):
        x = dense(p[f"Dense_{i}"], x)
        if ln and not ln_after_act:
            x = layer_norm(p[f"LayerNorm_{i}"], x)
        x = activation(x)
        if ln and ln_after_act:
            x = layer_norm(p[f"LayerNorm_{i}"], x)
    if has_output:
        x = dense(p[f"Dense_{n_hidden}"], x)
    return x


def residual_mlp_apply(p: Dict, x, n_blocks: int, activation, ln: bool):
    """ResidualMLP.__call__ (flowrl/module/mlp.py:77-98) with ResidualLinear (mlp.py:46-64), dropout off, equal block
    widths (residual.shape == x.shape, so no projection Dense): x = Dense_0(x); per block x = x + Dense_1(act(Dense_0([LN](x))));
    out = Dense_1(act(x)).  Flax names: Dense_0, ResidualLinear_i/{LayerNorm_0, Dense_0, Dense_1}, Dense_1."""
    x = dense(p["Dense_0"], x)
    for i in range(n_blocks):
        b = p[f"ResidualLinear_{i}"]
        y = layer_norm(b["LayerNorm_0"], x) if ln else x
        y = activation(dense(b["Dense_0"], y))
        x = x + dense(b["Dense_1"], y)
    return dense(p["Dense_1"], activation(x))


def simba_apply(p: Dict, x, n_blocks: int, activation):
    """Simba.__call__ (flowrl/module/simba.py:42-63) with SimbaBlock (simba.py:15-26): x = Dense_0(x); per block
    x = x + Dense_1(act(Dense_0(LN(x)))); out = Dense_1(LayerNorm_0(x)) — no activation before the output layer."""
    x = dense(p["Dense_0"], x)
    for i in range(n_blocks):
        b = p[f"SimbaBlock_{i}"]
        y = activation(dense(b["Dense_0"], layer_norm(b["LayerNorm_0"], x)))
        x = x + dense(b["Dense_1"], y)
    return dense(p["Dense_1"], layer_norm(p["LayerNorm_0"], x))


@dataclass
class BackboneSpec:
    """Static description of a DDPMBackbone / FlowBackbone instance."""
    hidden_dims: Sequence[int]
    x_dim: int
    activation: str = "mish"
    time_embedding: Optional[str] = "positional"   # "positional" | "learnable_fourier" | None (raw t)
    time_dim: int = 64
    cond_hidden: Optional[Sequence[int]] = None     # cond_embedding MLP hidden dims (dacer.py:263-266) or None
    predictor_name: str = "noise_predictor"         # "vel_predictor" for FlowBackbone
    layer_norm: bool = False                        # MLP(layer_norm=True) predictor (mlp.py:27-28)
    ln_after_act: bool = False                      # FQL's MLP: Dense -> act -> LayerNorm (fql/network.py:26-30)
    backbone: str = "mlp"                           # "mlp" | "residual" (ResidualMLP, mlp.py:67-98) | "simba" (simba.py:28-63);
                                                    # for the residual classes hidden_dims lists the BLOCKS
    multiplier: int = 4
    bf16_operands: bool = False                     # emulate the bf16-MFMA variant: predictor Dense operands rounded to
                                                    # bf16 (RNE), products / sums / biases / time slice / activations exact


def bf16_round(a):
    """Round to the nearest bfloat16 (ties to even), returned in a's dtype."""
    a = np.asarray(a)
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    u = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).astype(np.uint32)
    return u.view(np.float32).astype(a.dtype)


def _predictor_bf16(spec, p, x, time, condition, act):
    """The predictor MLP as the bf16 kernel computes it: layer inputs and kernels rounded to bf16, except the time
    slice of Dense_0, which the kernel folds (in f32) into the per-step bias table."""
    A, Dp = x.shape[-1], time.shape[-1]
    W0 = p["Dense_0"]["kernel"]
    z = bf16_round(x) @ bf16_round(W0[:A]) + time @ W0[A:A + Dp] + p["Dense_0"]["bias"]
    if condition is not None:
        z = z + bf16_round(condition) @ bf16_round(W0[A + Dp:])
    n = len(spec.hidden_dims)
    h = act(z)
    for i in range(1, n):
        h = act(bf16_round(h) @ bf16_round(p[f"Dense_{i}"]["kernel"]) + p[f"Dense_{i}"]["bias"])
    return bf16_round(h) @ bf16_round(p[f"Dense_{n}"]["kernel"]) + p[f"Dense_{n}"]["bias"]


def backbone_apply(spec: BackboneSpec, params: Dict, x, time, condition=None):
    """DDPMBackbone.__call__ (ddpm.py:83-95) == FlowBackbone.__call__ (cnf.py:30-42)."""
    act = get_activation(spec.activation)
    if spec.time_embedding is not None:
        if spec.time_embedding == "positional":
            time = positional_embedding(time, spec.time_dim)
        else:
            time = learnable_fourier_embedding(time, params["time_embedding"]["kernel"])
        # inline MLP(hidden_dims=[D, D], activation=mish), no output layer
        time = mlp_apply(params["MLP_0"], time, 2, mish, has_output=False)
    if spec.cond_hidden is not None:
        condition = mlp_apply(params["cond_embedding"], condition, len(spec.cond_hidden), act, has_output=False)
    if spec.bf16_operands:
        return _predictor_bf16(spec, params[spec.predictor_name], x, time, condition, act)
    if condition is not None:
        inputs = np.concatenate([x, time, condition], axis=-1)
    else:
        inputs = np.concatenate([x, time], axis=-1)
    if spec.backbone == "residual":
        return residual_mlp_apply(params[spec.predictor_name], inputs, len(spec.hidden_dims), act, spec.layer_norm)
    if spec.backbone == "simba":
        return simba_apply(params[spec.predictor_name], inputs, len(spec.hidden_dims), act)
    return mlp_apply(params[spec.predictor_name], inputs, len(spec.hidden_dims), act, has_output=True, ln=spec.layer_norm,
                     ln_after_act=spec.ln_after_act)


# --------------------------------------------------------------------------- #
# training-side half: forward process, one-step sample, losses
# --------------------------------------------------------------------------- #


def ddpm_add_noise(tables, x0, t, eps):
    """DDPM.add_noise given its draws (ddpm.py:170): t int (..., 1) in [1, steps]."""
    _, _, ah, _ = tables
    a = ah[t]
    return np.sqrt(a) * x0 + np.sqrt(1 - a) * eps


def continuous_add_noise(noise_schedule: str, x0, t, eps):
    """ContinuousDDPM.add_noise given its draws (continuous_ddpm.py:113-114)."""
    alpha, sigma = (cosine_noise_schedule if noise_schedule == "cosine" else linear_noise_schedule)(t)
    return alpha * x0 + sigma * eps


def linear_interpolation(x0, x1, t):
    """ContinuousNormalizingFlow.linear_interpolation given its draw (cnf.py:88-89): (xt, vel)."""
    return (1 - t) * x0 + t * x1, x1 - x0


def ddpm_onestep_update(tables, xt, eps_theta, t, num_samples, noise, *, solver="ddpm", clip_sampler=False, x_min=None, x_max=None):
    """The update half of DDPM.onestep_sample (ddpm.py:242-274) given the network output and the noise draw:
    returns (repeated_xt_1 (B, N, A), repeated_t_1 (B, N, 1))."""
    _, al, ah, _ = tables
    B = xt.shape[0]
    t_1 = t - 1
    repeated_t = t.repeat(num_samples, axis=0).reshape(B, num_samples, -1)
    repeated_t_1 = t_1.repeat(num_samples, axis=0).reshape(B, num_samples, -1)
    if solver == "ddpm":
        x0_hat = 1 / np.sqrt(ah[t]) * (xt - np.sqrt(1 - ah[t]) * eps_theta)
        if clip_sampler:
            x0_hat = _clip(x0_hat, x_min, x_max)
        xt_1 = 1 / (1 - ah[t]) * (np.sqrt(ah[t - 1]) * (1 - al[t]) * x0_hat + np.sqrt(al[t]) * (1 - ah[t - 1]) * xt)
        rep = xt_1.repeat(num_samples, axis=0).reshape(B, num_samples, -1)
        std_t = np.sqrt(1 - al[repeated_t])
        rep = rep + (repeated_t > 1) * std_t * noise
    elif solver == "ddim":
        a1 = 1 / np.sqrt(al[t])
        a2 = np.sqrt(1 - ah[t])
        a3 = np.sqrt(1 - ah[t - 1])
        xt_1 = a1 * (xt - a2 * eps_theta) + a3 * eps_theta
        rep = xt_1.repeat(num_samples, axis=0).reshape(B, num_samples, -1)
        if clip_sampler:
            rep = _clip(rep, x_min, x_max)
    else:
        raise NotImplementedError(solver)
    return rep, repeated_t_1


def mse(pred, target):
    """((pred - target) ** 2).mean() and its gradient w.r.t. pred (ddpm.py:294, continuous_ddpm.py:242, cnf.py:163)."""
    d = pred - target
    return (d ** 2).mean(), 2.0 * d / d.size


# --------------------------------------------------------------------------- #
# samplers
# --------------------------------------------------------------------------- #


def _clip(x, lo, hi):
    return np.clip(x, np.asarray(lo, dtype=x.dtype), np.asarray(hi, dtype=x.dtype))


def ddpm_sample(spec: BackboneSpec, params: Dict, tables, xT, condition, noise, *, steps: int,
                solver: str = "ddpm", clip_sampler: bool = False, x_min=None, x_max=None):
    """DDPM.sample — ddpm.py:173-221.  ``noise[j]`` is the N(0,I) draw of loop iteration j
    (t = T - j); it is consumed at every iteration (times 0 at t == 1), exactly like
    ``jax.random.normal(key_, xt.shape)`` at ddpm.py:206.  Returns (x0, (xt_hist, eps_hist))."""
    dt = xT.dtype
    betas, alphas, alpha_hats, postvars = [np.asarray(a, dtype=dt) for a in tables]
    xt = xT
    xt_hist, eps_hist = [], []
    one = np.ones((), dtype=dt)
    for j, t in enumerate(range(steps, 0, -1)):
        input_t = np.full((*xT.shape[:-1], 1), t, dtype=np.int32).astype(dt)   # t_proto * t, promoted to float
        eps = backbone_apply(spec, params, xt, input_t, condition)
        if solver == "ddpm":
            x0_hat = one / np.sqrt(alpha_hats[t]) * (xt - np.sqrt(one - alpha_hats[t]) * eps)
            if clip_sampler:
                x0_hat = _clip(x0_hat, x_min, x_max)
            xt_1 = one / (one - alpha_hats[t]) * (np.sqrt(alpha_hats[t - 1]) * (one - alphas[t]) * x0_hat
                                                  + np.sqrt(alphas[t]) * (one - alpha_hats[t - 1]) * xt)
            std_t = np.sqrt(postvars[t])
            xt_1 = xt_1 + np.asarray(float(t > 1), dtype=dt) * std_t * noise[j]
        elif solver == "ddim":
            a1 = one / np.sqrt(alphas[t])
            a2 = np.sqrt(one - alpha_hats[t])
            a3 = np.sqrt(one - alpha_hats[t - 1])
            xt_1 = a1 * (xt - a2 * eps) + a3 * eps
            if clip_sampler:
                xt_1 = _clip(xt_1, x_min, x_max)
        else:
            raise NotImplementedError(f"Unsupported solver {solver}")
        xt_hist.append(xt)
        eps_hist.append(eps)
        xt = xt_1.astype(dt)
    return xt, (np.stack(xt_hist), np.stack(eps_hist))


def continuous_ddpm_sample(spec: BackboneSpec, params: Dict, tables, xT, condition, noise, *, steps: int,
                           solver: str = "ddpm", clip_sampler: bool = False, x_min=None, x_max=None,
                           noise_schedule: str = "cosine", noise_schedule_params: Optional[dict] = None):
    """ContinuousDDPM.sample — continuous_ddpm.py:117-222 (solvers ddpm / ddim / dpm2 / dpm++2m).
    ``tables`` = (ts, alpha_hats, alphas, betas) from :func:`continuous_tables`; ``noise_schedule`` names the
    (alpha, sigma)(t) function dpm2 evaluates at the interval midpoints (continuous_ddpm.py:145-147)."""
    dt = xT.dtype
    ts, alpha_hats, alphas, betas = [np.asarray(a, dtype=dt) for a in tables]
    xt = xT
    xt_hist, eps_hist = [], []
    one = np.ones((), dtype=dt)
    if solver == "dpm++2m":      # continuous_ddpm.py:141-142
        lambdas = np.log(np.sqrt(alpha_hats) / np.sqrt(np.maximum(one - alpha_hats, np.asarray(1e-12, dtype=dt))))
    if solver == "dpm2":         # continuous_ddpm.py:145-147
        fn = {"cosine": cosine_noise_schedule, "linear": linear_noise_schedule}[noise_schedule]
        midpoints = ((ts[:-1] + ts[1:]) / np.asarray(2, dtype=dt)).astype(dt)
        mid_alphas, mid_sigmas = [np.asarray(v, dtype=dt) for v in fn(midpoints, **dict(noise_schedule_params or {}))]
    prev_x0_hat = np.zeros_like(xT)          # scan carry, continuous_ddpm.py:219
    prev_h = np.zeros((), dtype=dt)

    def net(x, tval):
        return backbone_apply(spec, params, x, np.ones((*xT.shape[:-1], 1), dtype=dt) * tval, condition)

    for j, i in enumerate(range(steps, 0, -1)):
        eps = net(xt, ts[i])
        x0_hat = (xt - np.sqrt(one - alpha_hats[i]) * eps) / np.sqrt(alpha_hats[i])
        if clip_sampler:
            x0_hat = _clip(x0_hat, x_min, x_max)
        h = np.zeros((), dtype=dt)
        if solver == "ddpm":
            c1 = np.sqrt(alpha_hats[i - 1]) * betas[i] / (one - alpha_hats[i])
            c2 = np.sqrt(alphas[i]) * (one - alpha_hats[i - 1]) / (one - alpha_hats[i])
            xt_1 = c1 * x0_hat + c2 * xt
            xt_1 = xt_1 + np.asarray(float(i > 1), dtype=dt) * np.sqrt(betas[i]) * noise[j]
        elif solver == "ddim":
            xt_1 = np.sqrt(alpha_hats[i - 1]) * x0_hat + np.sqrt(one - alpha_hats[i - 1]) * eps
        elif solver == "dpm2":               # continuous_ddpm.py:179-196: midpoint method, 2 network evaluations
            x_s = mid_alphas[i - 1] * x0_hat + mid_sigmas[i - 1] * eps
            eps_s = net(x_s.astype(dt), midpoints[i - 1])
            x0c = (xt - np.sqrt(one - alpha_hats[i]) * eps_s) / np.sqrt(alpha_hats[i])
            if clip_sampler:
                x0c = _clip(x0c, x_min, x_max)
            xt_1 = np.sqrt(alpha_hats[i - 1]) * x0c + np.sqrt(one - alpha_hats[i - 1]) * eps_s
        elif solver == "dpm++2m":            # continuous_ddpm.py:198-212: multistep, carries (x0_hat, h)
            h = (lambdas[i - 1] - lambdas[i]).astype(dt)
            sigma_curr, sigma_next = np.sqrt(one - alpha_hats[i]), np.sqrt(one - alpha_hats[i - 1])
            alpha_next = np.sqrt(alpha_hats[i - 1])
            r = prev_h / np.maximum(h, np.asarray(1e-8, dtype=dt))
            if prev_h > 1e-8:
                half_over_r = np.asarray(0.5, dtype=dt) / r
                d = (one + half_over_r) * x0_hat - half_over_r * prev_x0_hat
            else:
                d = x0_hat
            xt_1 = (sigma_next / sigma_curr) * xt - alpha_next * np.expm1(-h) * d
        else:
            raise NotImplementedError(f"Unsupported solver: {solver}")
        xt_hist.append(xt)
        eps_hist.append(eps)
        prev_x0_hat, prev_h = x0_hat, h
        xt = xt_1.astype(dt)
    return xt, (np.stack(xt_hist), np.stack(eps_hist))


def cnf_sample(spec: BackboneSpec, params: Dict, x0, condition, *, steps: int,
               clip_sampler: bool = False, x_min=None, x_max=None):
    """ContinuousNormalizingFlow.sample / _ode_step — cnf.py:92-141.  Euler, t = k / steps."""
    dt = x0.dtype
    xt = x0
    xt_hist, vel_hist = [], []
    tgrid = (np.arange(steps).astype(dt) / np.asarray(steps, dtype=dt)).astype(dt)   # step2t, cnf.py:80-83
    for k in range(steps):
        input_t = np.ones((*x0.shape[:-1], 1), dtype=np.int32).astype(dt) * tgrid[k]
        vel = backbone_apply(spec, params, xt, input_t, condition)
        x_next = xt + vel / np.asarray(steps, dtype=dt)                                # cnf.py:108
        if clip_sampler:
            x_next = _clip(x_next, x_min, x_max)
        xt_hist.append(xt)
        vel_hist.append(vel)
        xt = x_next.astype(dt)
    return xt, (np.stack(xt_hist), np.stack(vel_hist))


# --------------------------------------------------------------------------- #
# critic scoring + best-of-N selection — flowrl/module/critic.py:154-194, flowrl/agent/offline/dql.py:151-161
# --------------------------------------------------------------------------- #


def critic_apply(params: Dict, obs, action, n_hidden: int, activation: str):
    """Ensemblize(ScalarCritic(MLP)): every leaf carries the ensemble axis first (nn.vmap, variable_axes params = 0,
    in_axes=None: all members see the same input).  Returns (E, ..., 1)."""
    act = get_activation(activation)
    x = action if obs is None else np.concatenate([obs, action], axis=-1)      # critic.py:185-188
    out = []
    E = params["base"]["Dense_0"]["kernel"].shape[0]
    for e in range(E):
        h = x
        for i in range(n_hidden):
            d = params["base"]["backbone"][f"Dense_{i}"]
            h = act(h @ d["kernel"][e] + d["bias"][e])
        d = params["base"]["Dense_0"]
        out.append(h @ d["kernel"][e] + d["bias"][e])
    return np.stack(out)


def select_best(params: Dict, obs, actions, n_hidden: int, activation: str, temperature=0.0, gumbel=None):
    """dql.py:145,151-161 after the sampler: obs (B, S), actions (B, N, A).  Returns (chosen (B, A), idx (B,), qs (B, N)).
    temperature > 0: jax.random.categorical(key, logits) == argmax(logits + Gumbel(key)); the noise is an input."""
    B, N = actions.shape[0], actions.shape[1]
    obs_repeat = None if obs is None else np.repeat(obs[:, None, :], N, axis=1)
    qs = critic_apply(params, obs_repeat, actions, n_hidden, activation)       # (E, B, N, 1)
    qs = qs.min(axis=0).reshape(B, N)
    if temperature is None or temperature <= 0.0:
        idx = qs.argmax(axis=-1)
    else:
        idx = (qs / np.asarray(1e-6 + temperature, dtype=qs.dtype) + gumbel).argmax(axis=-1)
    return actions[np.arange(B), idx], idx, qs


def make_critic_params(obs_dim: int, act_dim: int, hidden_dims, ensemble: int, seed: int, dtype=np.float32, bias_std=0.1) -> Dict:
    rng = np.random.default_rng(seed)
    dims = [obs_dim + act_dim, *hidden_dims]

    def stacked(n_in, n_out):
        ms = [_dense_params(rng, n_in, n_out, dtype, bias_std) for _ in range(ensemble)]
        return {"kernel": np.stack([m["kernel"] for m in ms]), "bias": np.stack([m["bias"] for m in ms])}

    return {"base": {"backbone": {f"Dense_{i}": stacked(dims[i], dims[i + 1]) for i in range(len(hidden_dims))},
                     "Dense_0": stacked(dims[-1], 1)}}


# --------------------------------------------------------------------------- #
# synthetic parameters (SURVEY.md section 8d "Synthetic values & seeds")
# --------------------------------------------------------------------------- #


def _orthogonal(rng, n_in, n_out, gain):
    # flowrl/module/initialization.py:9-12 -> nn.initializers.orthogonal(sqrt(2))
    a = rng.standard_normal((max(n_in, n_out), min(n_in, n_out)))
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diag(r))
    if n_in < n_out:
        q = q.T
    return (gain * q[:n_in, :n_out])


def _dense_params(rng, n_in, n_out, dtype, bias_std=0.1):
    return {
        "kernel": _orthogonal(rng, n_in, n_out, math.sqrt(2.0)).astype(dtype),
        "bias": (bias_std * rng.standard_normal((n_out,))).astype(dtype),   # non-zero on purpose: zero biases hide bias bugs
    }


def make_params(spec: BackboneSpec, cond_dim: int, seed: int, dtype=np.float32, cond_in_dim: Optional[int] = None) -> Dict:
    """Random parameter tree in the Flax layout of DDPMBackbone / FlowBackbone (ddpm.py:70-95).
    ``cond_dim`` is the feature width that reaches the predictor; ``cond_in_dim`` the raw obs width
    when a cond_embedding MLP is present."""
    rng = np.random.default_rng(seed)
    p: Dict = {}
    d_time = spec.time_dim if spec.time_embedding is not None else 1
    if spec.time_embedding is not None:
        p["MLP_0"] = {
            "Dense_0": _dense_params(rng, spec.time_dim, spec.time_dim, dtype),
            "Dense_1": _dense_params(rng, spec.time_dim, spec.time_dim, dtype),
        }
        if spec.time_embedding == "learnable_fourier":
            p["time_embedding"] = {"kernel": (0.2 * rng.standard_normal((spec.time_dim // 2, 1))).astype(dtype)}
    if spec.cond_hidden is not None:
        dims = [cond_in_dim] + list(spec.cond_hidden)
        p["cond_embedding"] = {f"Dense_{i}": _dense_params(rng, dims[i], dims[i + 1], dtype) for i in range(len(spec.cond_hidden))}
        cond_dim = spec.cond_hidden[-1]
    if spec.backbone != "mlp":
        h, m = int(spec.hidden_dims[0]), int(spec.multiplier)
        block = "SimbaBlock" if spec.backbone == "simba" else "ResidualLinear"
        ln = spec.layer_norm or spec.backbone == "simba"
        lnp = lambda: {"scale": (1.0 + 0.2 * rng.standard_normal(h)).astype(dtype), "bias": (0.1 * rng.standard_normal(h)).astype(dtype)}
        q: Dict = {"Dense_0": _dense_params(rng, spec.x_dim + d_time + cond_dim, h, dtype)}
        for i in range(len(spec.hidden_dims)):
            q[f"{block}_{i}"] = {"Dense_0": _dense_params(rng, h, m * h, dtype), "Dense_1": _dense_params(rng, m * h, h, dtype)}
            if ln:
                q[f"{block}_{i}"]["LayerNorm_0"] = lnp()
        if spec.backbone == "simba":
            q["LayerNorm_0"] = lnp()
        q["Dense_1"] = _dense_params(rng, h, spec.x_dim, dtype)
        p[spec.predictor_name] = q
        return p
    dims = [spec.x_dim + d_time + cond_dim] + list(spec.hidden_dims) + [spec.x_dim]
    p[spec.predictor_name] = {f"Dense_{i}": _dense_params(rng, dims[i], dims[i + 1], dtype) for i in range(len(dims) - 1)}
    if spec.layer_norm:   # drawn AFTER everything else so that specs without LayerNorm keep their parameter streams
        for i, h in enumerate(spec.hidden_dims):
            p[spec.predictor_name][f"LayerNorm_{i}"] = {"scale": (1.0 + 0.2 * rng.standard_normal(h)).astype(dtype),
                                                        "bias": (0.1 * rng.standard_normal(h)).astype(dtype)}
    return p


def cast_tree(tree, dtype):
    if isinstance(tree, dict):
        return {k: cast_tree(v, dtype) for k, v in tree.items()}
    return np.asarray(tree, dtype=dtype)


def flatten_tree(tree, prefix=""):
    out = {}
    for k, v in tree.items():
        key = f"{prefix}/{k}" if prefix else k
        if isinstance(v, dict):
            out.update(flatten_tree(v, key))
        else:
            out[key] = v
    return out


def unflatten_tree(flat):
    out: Dict = {}
    for key, v in flat.items():
        node = out
        parts = key.split("/")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = v
    return out
