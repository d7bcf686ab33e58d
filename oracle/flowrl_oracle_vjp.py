"""CPU oracle for the backprop-through-chain VJP (DQL-style actor update), numpy, manual backprop.

TEST INFRASTRUCTURE ONLY (same rules as flowrl_oracle.py).  PARITY UNPINNED by the reference: it has no
tests; what it does is ``jax.grad`` through the unrolled scan of ``DDPM.sample(training=True, params=p)``
(flowrl/agent/offline/dql.py:93-123, flowrl/module/model.py:78-86, flowrl/flow/ddpm.py:186-219).  This file
restates that derivative by hand; tests pin it with central finite differences in float64 and with torch
autograd over an independent restatement (tests/test_oracle_vjp.py).

    loss = sum(g * x0(theta))      =>      returns d loss / d theta   (theta = the Flax parameter tree)
"""
from __future__ import annotations

from typing import Dict

import numpy as np

import flowrl_oracle as orc


# ---- activations and their derivatives -------------------------------------------------------------
def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def act_grad(name: str, x):
    if name == "mish":
        sp = np.logaddexp(x, 0.0)
        t = np.tanh(sp)
        return t + x * (1.0 - t * t) * _sigmoid(x)
    if name == "silu":
        s = _sigmoid(x)
        return s * (1.0 + x * (1.0 - s))
    if name == "relu":
        return (x > 0).astype(x.dtype)
    if name == "tanh":
        return 1.0 - np.tanh(x) ** 2
    if name == "elu":
        return np.where(x > 0, 1.0, np.exp(np.minimum(x, 0.0)))
    if name == "gelu_tanh":
        c, k = np.sqrt(2.0 / np.pi), 0.044715
        u = c * (x + k * x ** 3)
        th = np.tanh(u)
        return 0.5 * (1.0 + th) + 0.5 * x * (1.0 - th * th) * c * (1.0 + 3.0 * k * x * x)
    raise KeyError(name)


# ---- network forward with cache / backward ------------------------------------------------------------
def _mlp_fwd(p, x, n_hidden, act_name, has_output):
    act = orc.get_activation(act_name)
    cache = []
    for i in range(n_hidden):
        z = x @ p[f"Dense_{i}"]["kernel"] + p[f"Dense_{i}"]["bias"]
        cache.append((x, z))
        x = act(z)
    if has_output:
        cache.append((x, None))
        x = x @ p[f"Dense_{n_hidden}"]["kernel"] + p[f"Dense_{n_hidden}"]["bias"]
    return x, cache


def _mlp_bwd(p, cache, gout, n_hidden, act_name, has_output, grads):
    """Accumulates parameter grads into ``grads`` (same tree layout as p); returns grad wrt the MLP input."""
    g = gout
    if has_output:
        xin, _ = cache[n_hidden]
        d = grads.setdefault(f"Dense_{n_hidden}", {"kernel": 0.0, "bias": 0.0})
        d["kernel"] = d["kernel"] + xin.reshape(-1, xin.shape[-1]).T @ g.reshape(-1, g.shape[-1])
        d["bias"] = d["bias"] + g.reshape(-1, g.shape[-1]).sum(0)
        g = g @ p[f"Dense_{n_hidden}"]["kernel"].T
    for i in reversed(range(n_hidden)):
        xin, z = cache[i]
        g = g * act_grad(act_name, z)
        d = grads.setdefault(f"Dense_{i}", {"kernel": 0.0, "bias": 0.0})
        d["kernel"] = d["kernel"] + xin.reshape(-1, xin.shape[-1]).T @ g.reshape(-1, g.shape[-1])
        d["bias"] = d["bias"] + g.reshape(-1, g.shape[-1]).sum(0)
        g = g @ p[f"Dense_{i}"]["kernel"].T
    return g


# ---- LayerNorm / residual predictors (mlp.py:26-29, fql/network.py:26-30, mlp.py:37-98, simba.py:10-63) ----------------
def _ln_fwd(p, x, eps=1e-6):
    """flax.linen.LayerNorm as orc.layer_norm computes it, keeping (xhat, rstd) for the backward."""
    mean = x.mean(axis=-1, keepdims=True)
    var = np.maximum((x * x).mean(axis=-1, keepdims=True) - mean * mean, 0)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean) * rstd
    return xhat * p["scale"] + p["bias"], (xhat, rstd)


def _ln_bwd(p, cache, dy, grads):
    xhat, rstd = cache
    grads["scale"] = grads.get("scale", 0.0) + (dy * xhat).reshape(-1, xhat.shape[-1]).sum(0)
    grads["bias"] = grads.get("bias", 0.0) + dy.reshape(-1, xhat.shape[-1]).sum(0)
    dxh = dy * p["scale"]
    return rstd * (dxh - dxh.mean(axis=-1, keepdims=True) - xhat * (dxh * xhat).mean(axis=-1, keepdims=True))


def _dense_bwd(p, xin, g, grads):
    grads["kernel"] = grads.get("kernel", 0.0) + xin.reshape(-1, xin.shape[-1]).T @ g.reshape(-1, g.shape[-1])
    grads["bias"] = grads.get("bias", 0.0) + g.reshape(-1, g.shape[-1]).sum(0)
    return g @ p["kernel"].T


def _pred_fwd(spec, p, x):
    """The predictor (orc.mlp_apply / residual_mlp_apply / simba_apply) keeping what its backward needs."""
    act = orc.get_activation(spec.activation)
    n = len(spec.hidden_dims)
    tape = []
    if spec.backbone == "mlp":
        for i in range(n):
            z = orc.dense(p[f"Dense_{i}"], x)
            ent = {"xin": x, "z": z}
            if spec.layer_norm and not spec.ln_after_act:
                nz, ent["ln"] = _ln_fwd(p[f"LayerNorm_{i}"], z)
                ent["n"] = nz
                x = act(nz)
            elif spec.layer_norm:
                x, ent["ln"] = _ln_fwd(p[f"LayerNorm_{i}"], act(z))
            else:
                x = act(z)
            tape.append(ent)
        tape.append({"xin": x})
        return orc.dense(p[f"Dense_{n}"], x), tape
    simba = spec.backbone == "simba"
    ln = spec.layer_norm or simba
    bname = "SimbaBlock" if simba else "ResidualLinear"
    ent0 = {"xin": x}
    x = orc.dense(p["Dense_0"], x)
    tape.append(ent0)
    for i in range(n):
        b = p[f"{bname}_{i}"]
        ent = {}
        y = x
        if ln:
            y, ent["ln"] = _ln_fwd(b["LayerNorm_0"], x)
        ent["y"] = y
        e = orc.dense(b["Dense_0"], y)
        ent["e"] = e
        a = act(e)
        ent["a"] = a
        x = x + orc.dense(b["Dense_1"], a)
        tape.append(ent)
    ent = {"x": x}
    if simba:
        hf, ent["ln"] = _ln_fwd(p["LayerNorm_0"], x)
    else:
        hf = act(x)
    ent["hf"] = hf
    tape.append(ent)
    return orc.dense(p["Dense_1"], hf), tape


def _pred_bwd(spec, p, tape, gout, grads):
    n = len(spec.hidden_dims)
    if spec.backbone == "mlp":
        g = _dense_bwd(p[f"Dense_{n}"], tape[n]["xin"], gout, grads.setdefault(f"Dense_{n}", {}))
        for i in reversed(range(n)):
            ent = tape[i]
            if spec.layer_norm and not spec.ln_after_act:
                g = g * act_grad(spec.activation, ent["n"])
                g = _ln_bwd(p[f"LayerNorm_{i}"], ent["ln"], g, grads.setdefault(f"LayerNorm_{i}", {}))
            elif spec.layer_norm:
                g = _ln_bwd(p[f"LayerNorm_{i}"], ent["ln"], g, grads.setdefault(f"LayerNorm_{i}", {}))
                g = g * act_grad(spec.activation, ent["z"])
            else:
                g = g * act_grad(spec.activation, ent["z"])
            g = _dense_bwd(p[f"Dense_{i}"], ent["xin"], g, grads.setdefault(f"Dense_{i}", {}))
        return g
    simba = spec.backbone == "simba"
    ln = spec.layer_norm or simba
    bname = "SimbaBlock" if simba else "ResidualLinear"
    head = tape[n + 1]
    g = _dense_bwd(p["Dense_1"], head["hf"], gout, grads.setdefault("Dense_1", {}))
    if simba:
        g = _ln_bwd(p["LayerNorm_0"], head["ln"], g, grads.setdefault("LayerNorm_0", {}))
    else:
        g = g * act_grad(spec.activation, head["x"])
    for i in reversed(range(n)):
        b, ent = p[f"{bname}_{i}"], tape[1 + i]
        gb = grads.setdefault(f"{bname}_{i}", {})
        ga = _dense_bwd(b["Dense_1"], ent["a"], g, gb.setdefault("Dense_1", {}))
        ge = ga * act_grad(spec.activation, ent["e"])
        gy = _dense_bwd(b["Dense_0"], ent["y"], ge, gb.setdefault("Dense_0", {}))
        if ln:
            gy = _ln_bwd(b["LayerNorm_0"], ent["ln"], gy, gb.setdefault("LayerNorm_0", {}))
        g = g + gy
    return _dense_bwd(p["Dense_0"], tape[0]["xin"], g, grads.setdefault("Dense_0", {}))


# ---- predictor with bf16 operands (emulation of csrc/sampler_bf16.hip + csrc/sampler_bwd_bf16.hip) ------------
# Every Dense contraction takes operands rounded to bf16 and sums exactly: layer inputs and kernels in the forward
# (orc._predictor_bf16), and in the backward dL/deps and the transposed kernels.  The kernel stashes z_l ONCE, rounded to
# bf16, and its consumers recompute from that: dL/dz_l = bf16(dL/dh_l * act'(bf16(z_l))), and the X operand of dW_{l+1} is
# bf16(act(bf16(z_l))) (the forward itself uses act of the unrounded z).  The time slice of Dense_0 (folded into the f32
# bias table by the kernel) and all biases stay exact.
def _pred_fwd_bf16(spec, p, x, tfeat, condition):
    bf = orc.bf16_round
    act = orc.get_activation(spec.activation)
    A, Dp = x.shape[-1], tfeat.shape[-1]
    n = len(spec.hidden_dims)
    W0 = p["Dense_0"]["kernel"]
    parts = [bf(x), tfeat] + ([bf(condition)] if condition is not None else [])
    xin = np.concatenate(parts, axis=-1)
    W0b = np.concatenate([bf(W0[:A]), W0[A:A + Dp], bf(W0[A + Dp:])], axis=0)
    cache = []
    z = xin @ W0b + p["Dense_0"]["bias"]
    cache.append((xin, z, W0b))
    for i in range(1, n):
        hb = bf(act(z))
        Wb = bf(p[f"Dense_{i}"]["kernel"])
        z = hb @ Wb + p[f"Dense_{i}"]["bias"]
        cache.append((hb, z, Wb))
    hb = bf(act(z))
    Wb = bf(p[f"Dense_{n}"]["kernel"])
    cache.append((hb, None, Wb))
    return hb @ Wb + p[f"Dense_{n}"]["bias"], cache


def _pred_bwd_bf16(spec, cache, gout, grads):
    bf = orc.bf16_round
    n = len(spec.hidden_dims)

    def acc(i, xin, g):
        d = grads.setdefault(f"Dense_{i}", {"kernel": 0.0, "bias": 0.0})
        d["kernel"] = d["kernel"] + xin.reshape(-1, xin.shape[-1]).T @ g.reshape(-1, g.shape[-1])
        d["bias"] = d["bias"] + g.reshape(-1, g.shape[-1]).sum(0)

    g = bf(gout)
    xin, _, Wb = cache[n]
    xin = bf(orc.get_activation(spec.activation)(bf(cache[n - 1][1])))
    acc(n, xin, g)
    g = g @ Wb.T
    for i in reversed(range(n)):
        xin, z, Wb = cache[i]
        g = bf(g * act_grad(spec.activation, bf(z)))
        if i > 0:      # dW_i's X operand: act of the STASHED (bf16) z_{i-1}
            xin = bf(orc.get_activation(spec.activation)(bf(cache[i - 1][1])))
        acc(i, xin, g)
        g = g @ Wb.T
    return g


def backbone_fwd(spec: orc.BackboneSpec, params: Dict, x, time, condition):
    """DDPMBackbone.__call__ (ddpm.py:83-95) keeping what the backward needs."""
    c = {"time_in": time}
    if spec.time_embedding is not None:
        if spec.time_embedding == "positional":
            temb = orc.positional_embedding(time, spec.time_dim)
        else:
            temb = orc.learnable_fourier_embedding(time, params["time_embedding"]["kernel"])
        tfeat, c["tmlp"] = _mlp_fwd(params["MLP_0"], temb, 2, "mish", False)
    else:
        tfeat = time
    if spec.cond_hidden is not None:
        condition, c["cmlp"] = _mlp_fwd(params["cond_embedding"], condition, len(spec.cond_hidden), spec.activation, False)
    c["dims"] = (x.shape[-1], tfeat.shape[-1])
    if spec.bf16_operands:
        out, c["pred"] = _pred_fwd_bf16(spec, params[spec.predictor_name], x, tfeat, condition)
        return out, c
    parts = [x, tfeat] + ([condition] if condition is not None else [])
    inputs = np.concatenate(parts, axis=-1)
    if spec.backbone != "mlp" or spec.layer_norm:
        out, c["pred"] = _pred_fwd(spec, params[spec.predictor_name], inputs)
        return out, c
    out, c["pred"] = _mlp_fwd(params[spec.predictor_name], inputs, len(spec.hidden_dims), spec.activation, True)
    return out, c


def backbone_bwd(spec: orc.BackboneSpec, params: Dict, c, gout, grads: Dict):
    if spec.bf16_operands:
        gin = _pred_bwd_bf16(spec, c["pred"], gout, grads.setdefault(spec.predictor_name, {}))
    elif spec.backbone != "mlp" or spec.layer_norm:
        gin = _pred_bwd(spec, params[spec.predictor_name], c["pred"], gout, grads.setdefault(spec.predictor_name, {}))
    else:
        gin = _mlp_bwd(params[spec.predictor_name], c["pred"], gout, len(spec.hidden_dims), spec.activation, True,
                       grads.setdefault(spec.predictor_name, {}))
    A, Dp = c["dims"]
    gx, gt, gc = gin[..., :A], gin[..., A:A + Dp], gin[..., A + Dp:]
    if spec.cond_hidden is not None:
        _mlp_bwd(params["cond_embedding"], c["cmlp"], gc, len(spec.cond_hidden), spec.activation, False,
                 grads.setdefault("cond_embedding", {}))
    if spec.time_embedding is not None:
        gtemb = _mlp_bwd(params["MLP_0"], c["tmlp"], gt, 2, "mish", False, grads.setdefault("MLP_0", {}))
        if spec.time_embedding == "learnable_fourier":
            # temb = [cos f, sin f], f = 2 pi t @ w.T   (time_embedding.py:23-27)
            w = params["time_embedding"]["kernel"]
            t = c["time_in"]
            f = 2 * np.pi * t @ w.T
            half = w.shape[0]
            gf = -np.sin(f) * gtemb[..., :half] + np.cos(f) * gtemb[..., half:]
            gw = (2 * np.pi * t.reshape(-1, t.shape[-1])).T @ gf.reshape(-1, half)     # (in_dim, D/2)
            d = grads.setdefault("time_embedding", {"kernel": 0.0})
            d["kernel"] = d["kernel"] + gw.T
    return gx


# ---- chain VJP --------------------------------------------------------------------------------------------
def _chain_vjp(spec, params, xT, condition, g, steps, step_fwd, xt_hist=None):
    """Shared driver: ``step_fwd(j, xt, eps) -> (x_next, bwd)`` with ``bwd(g_next) -> (g_eps, g_xt_direct)``;
    the time input of iteration j comes from ``step_fwd.tvals[j]``."""
    dt = np.float64
    params = orc.cast_tree(params, dt)
    xt = np.asarray(xT, dt)
    cond = None if condition is None else np.asarray(condition, dt)
    fw = []
    for j in range(steps):
        if xt_hist is not None:
            xt = np.asarray(xt_hist[j], dt)
        tin = np.full((*xt.shape[:-1], 1), float(step_fwd.tvals[j]), dtype=dt)
        eps, cache = backbone_fwd(spec, params, xt, tin, cond)
        xn, bwd = step_fwd(j, xt, eps)
        fw.append((cache, bwd))
        xt = xn
    x0 = xt
    grads: Dict = {}
    gx = np.asarray(g, dt)
    for j in reversed(range(steps)):
        cache, bwd = fw[j]
        g_eps, g_dir = bwd(gx)
        gx = g_dir + backbone_bwd(spec, params, cache, g_eps, grads)
    return x0, grads, gx


def _clip_mask(v, clip_sampler, x_min, x_max):
    if not clip_sampler:
        return v, np.ones_like(v)
    return np.clip(v, x_min, x_max), ((v >= x_min) & (v <= x_max)).astype(np.float64)


def ddpm_sample_vjp(spec: orc.BackboneSpec, params: Dict, tables, xT, condition, noise, g, *, steps: int,
                    solver: str = "ddpm", clip_sampler: bool = False, x_min=None, x_max=None, xt_hist=None):
    """d sum(g * x0) / d params and / d xT for DDPM.sample (ddpm.py:173-221), solvers ddpm and ddim.  float64 throughout.
    ``xt_hist`` (T, ..., A), optional: take the input of iteration j from it instead of from the recomputed chain —
    the product's VJP works from the forward call's history[0] the same way, so a reduced-precision variant can be
    checked on identical chain states."""
    dt = np.float64
    betas, alphas, alpha_hats, postvars = [np.asarray(a, dtype=dt) for a in tables]
    noise = None if noise is None else np.asarray(noise, dt)

    def step(j, xt, eps):
        t = steps - j
        ah, ah1, al = alpha_hats[t], alpha_hats[t - 1], alphas[t]
        if solver == "ddpm":                                   # ddpm.py:198-206
            c0, c1 = 1.0 / np.sqrt(ah), np.sqrt(1.0 - ah)
            x0h, mask = _clip_mask(c0 * (xt - c1 * eps), clip_sampler, x_min, x_max)
            c2, c3, c4 = 1.0 / (1.0 - ah), np.sqrt(ah1) * (1.0 - al), np.sqrt(al) * (1.0 - ah1)
            xn = c2 * (c3 * x0h + c4 * xt) + float(t > 1) * np.sqrt(postvars[t]) * noise[j]

            def bwd(gn):
                g0 = c2 * c3 * gn * mask
                return -c0 * c1 * g0, c2 * c4 * gn + c0 * g0
        elif solver == "ddim":                                 # ddpm.py:207-213: clips xt_1, prefactor 1/sqrt(alpha_t)
            a1, a2, a3 = 1.0 / np.sqrt(al), np.sqrt(1.0 - ah), np.sqrt(1.0 - ah1)
            xn, mask = _clip_mask(a1 * (xt - a2 * eps) + a3 * eps, clip_sampler, x_min, x_max)

            def bwd(gn):
                gp = gn * mask
                return gp * (a3 - a1 * a2), a1 * gp
        else:
            raise NotImplementedError(solver)
        return xn, bwd

    step.tvals = [float(steps - j) for j in range(steps)]
    return _chain_vjp(spec, params, xT, condition, g, steps, step, xt_hist)


def continuous_ddpm_sample_vjp(spec: orc.BackboneSpec, params: Dict, tables, xT, condition, noise, g, *, steps: int,
                               solver: str = "ddpm", clip_sampler: bool = False, x_min=None, x_max=None, xt_hist=None):
    """The same for ContinuousDDPM.sample (continuous_ddpm.py:159-177, solvers ddpm and ddim) — what DACER's actor loss
    differentiates (agent/online/dacer.py:141-162).  ``tables`` = (ts, alpha_hats, alphas, betas)."""
    dt = np.float64
    ts, alpha_hats, alphas, betas = [np.asarray(a, dtype=dt) for a in tables]
    noise = None if noise is None else np.asarray(noise, dt)

    def step(j, xt, eps):
        i = steps - j
        ah, ah1, al, be = alpha_hats[i], alpha_hats[i - 1], alphas[i], betas[i]
        c0, c1 = np.sqrt(ah), np.sqrt(1.0 - ah)
        x0h, mask = _clip_mask((xt - c1 * eps) / c0, clip_sampler, x_min, x_max)
        if solver == "ddpm":
            m1, m2 = np.sqrt(ah1) * be / (1.0 - ah), np.sqrt(al) * (1.0 - ah1) / (1.0 - ah)
            xn = m1 * x0h + m2 * xt + float(i > 1) * np.sqrt(be) * noise[j]

            def bwd(gn):
                g0 = m1 * gn * mask
                return -(c1 / c0) * g0, m2 * gn + g0 / c0
        elif solver == "ddim":
            m1, m2 = np.sqrt(ah1), np.sqrt(1.0 - ah1)
            xn = m1 * x0h + m2 * eps

            def bwd(gn):
                g0 = m1 * gn * mask
                return -(c1 / c0) * g0 + m2 * gn, g0 / c0
        else:
            raise NotImplementedError(solver)
        return xn, bwd

    step.tvals = [float(ts[steps - j]) for j in range(steps)]
    return _chain_vjp(spec, params, xT, condition, g, steps, step, xt_hist)


def cnf_sample_vjp(spec: orc.BackboneSpec, params: Dict, x0, condition, g, *, steps: int, clip_sampler: bool = False,
                   x_min=None, x_max=None, xt_hist=None):
    """The same for ContinuousNormalizingFlow.sample (cnf.py:92-141): x_next = clip(xt + vel / steps)."""
    def step(j, xt, vel):
        xn, mask = _clip_mask(xt + vel / float(steps), clip_sampler, x_min, x_max)

        def bwd(gn):
            gp = gn * mask
            return gp / float(steps), gp
        return xn, bwd

    # step2t(arange(steps)) in float32 (cnf.py:80-83), as the product computes it
    step.tvals = [float(np.float32(j) / np.float32(steps)) for j in range(steps)]
    return _chain_vjp(spec, params, x0, condition, g, steps, step, xt_hist)
