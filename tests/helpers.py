"""Shared test helpers: one configuration -> oracle inputs (numpy) and product inputs (torch)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

import flowrl_oracle as orc


# per-case parity margins of a GPU session (tests/conftest.py writes them to gpurun_out/parity_margins.json at session end;
# a copy is committed as profiles/rNN_parity_margins.json): how close to its bar each case sits
MARGINS: list = []


def record_margin(suite: str, case: str, build: str, **metrics):
    MARGINS.append({"suite": suite, "case": case, "build": build,
                    **{k: (float(v) if v is not None else None) for k, v in metrics.items()}})


@dataclass
class Case:
    name: str
    kind: str                      # "ddpm" | "cddpm" | "cnf"
    A: int
    S: int                         # cond dim (0 = none)
    T: int
    B: int
    hidden: tuple = (256, 256, 256)
    act: str = "mish"
    temb: Optional[str] = "positional"
    schedule: str = "vp"
    solver: str = "ddpm"
    clip: bool = True
    x_min: float = -1.0
    x_max: float = 1.0
    approx_postvar: bool = False
    particles: int = 1
    seed: int = 0
    cond_embed: bool = False       # cond_embedding MLP(128,128): S is then the RAW obs width
    bf16: bool = False             # bf16-MFMA variant (product) / bf16-operand emulation (oracle)
    tile: str = "auto"             # FrlDesc.tile_policy of the product run: auto | throughput | latency
    ln: bool = False               # MLP(layer_norm=True) predictor: Dense -> LayerNorm -> act (mlp.py:26-29)
    ln_post: bool = False          # with ln: FQL's order Dense -> act -> LayerNorm (fql/network.py:26-30)
    backbone: str = "mlp"          # "mlp" | "residual" (ResidualMLP) | "simba": `hidden` then lists the blocks

    @property
    def spec(self):
        return orc.BackboneSpec(hidden_dims=list(self.hidden), x_dim=self.A, activation=self.act,
                                time_embedding=self.temb, time_dim=64,
                                cond_hidden=[128, 128] if self.cond_embed else None,
                                predictor_name="vel_predictor" if self.kind == "cnf" else "noise_predictor",
                                bf16_operands=self.bf16, layer_norm=self.ln, ln_after_act=self.ln_post, backbone=self.backbone)


def make_inputs(c: Case):
    """numpy float32 inputs + params (Flax layout) for a case; condition is un-repeated (n_states, S)."""
    rng = np.random.default_rng(1000 + c.seed)
    params = orc.make_params(c.spec, c.S, seed=c.seed, cond_in_dim=c.S)
    n_states = c.B // c.particles
    assert n_states * c.particles == c.B
    xT = rng.standard_normal((c.B, c.A)).astype(np.float32)
    cond = rng.standard_normal((n_states, c.S)).astype(np.float32) if c.S > 0 else None
    noise = rng.standard_normal((c.T, c.B, c.A)).astype(np.float32)
    return params, xT, cond, noise


def oracle_run(c: Case, params, xT, cond, noise, dtype=np.float64):
    """Run the oracle in `dtype`; returns (x0, xt_hist, eps_hist)."""
    p = orc.cast_tree(params, dtype)
    x = xT.astype(dtype)
    cd = None if cond is None else np.repeat(cond.astype(dtype), c.particles, axis=0)
    nz = noise.astype(dtype)
    kw = dict(steps=c.T, clip_sampler=c.clip, x_min=c.x_min, x_max=c.x_max)
    if c.kind == "ddpm":
        tables = orc.ddpm_tables(c.T, c.schedule, approx_postvar=c.approx_postvar, dtype=np.float32)
        x0, (xh, eh) = orc.ddpm_sample(c.spec, p, [t.astype(dtype) for t in tables], x, cd, nz, solver=c.solver, **kw)
    elif c.kind == "cddpm":
        tables = orc.continuous_tables(c.T, c.schedule, dtype=np.float32)
        x0, (xh, eh) = orc.continuous_ddpm_sample(c.spec, p, [t.astype(dtype) for t in tables], x, cd, nz, solver=c.solver,
                                                  noise_schedule=c.schedule, **kw)
    else:
        x0, (xh, eh) = orc.cnf_sample(c.spec, p, x, cd, **kw)
    return x0, xh, eh


def permuted_hidden_units(c: Case, params, seed: int):
    """The SAME network with the hidden units of every predictor layer renumbered (columns of Dense_l, its bias and
    LayerNorm rows, rows of Dense_{l+1}): mathematically identical outputs, another float32 summation order."""
    rng = np.random.default_rng(seed)
    name = c.spec.predictor_name
    if c.backbone != "mlp":
        return permuted_residual_units(c, params, rng)
    out = {k: (dict(v) if isinstance(v, dict) else v) for k, v in params.items()}
    pp = {k: dict(v) for k, v in params[name].items()}
    L = len(c.hidden)
    for l in range(L):
        perm = rng.permutation(c.hidden[l])
        pp[f"Dense_{l}"]["kernel"] = pp[f"Dense_{l}"]["kernel"][:, perm]
        pp[f"Dense_{l}"]["bias"] = pp[f"Dense_{l}"]["bias"][perm]
        if c.ln:
            pp[f"LayerNorm_{l}"]["scale"] = pp[f"LayerNorm_{l}"]["scale"][perm]
            pp[f"LayerNorm_{l}"]["bias"] = pp[f"LayerNorm_{l}"]["bias"][perm]
        pp[f"Dense_{l + 1}"]["kernel"] = pp[f"Dense_{l + 1}"]["kernel"][perm, :]
    out[name] = pp
    return out


def permuted_residual_units(c: Case, params, rng):
    """Residual classes: renumber the 4H expansion units of every block (columns of its Dense_0, rows of its Dense_1) and
    the trunk's H units (every tensor that reads or writes the residual stream)."""
    name = c.spec.predictor_name
    out = {k: (dict(v) if isinstance(v, dict) else v) for k, v in params.items()}
    pp = {k: ({kk: (dict(vv) if isinstance(vv, dict) else vv) for kk, vv in v.items()}) for k, v in params[name].items()}
    h = c.hidden[0]
    trunk = rng.permutation(h)
    pp["Dense_0"]["kernel"] = pp["Dense_0"]["kernel"][:, trunk]
    pp["Dense_0"]["bias"] = pp["Dense_0"]["bias"][trunk]
    pp["Dense_1"]["kernel"] = pp["Dense_1"]["kernel"][trunk, :]
    if "LayerNorm_0" in pp:
        pp["LayerNorm_0"] = {k: v[trunk] for k, v in pp["LayerNorm_0"].items()}
    for key in list(pp):
        if "_" in key and key.split("_")[0] in ("ResidualLinear", "SimbaBlock"):
            b = pp[key]
            wide = rng.permutation(b["Dense_0"]["kernel"].shape[1])
            b["Dense_0"] = {"kernel": b["Dense_0"]["kernel"][trunk][:, wide], "bias": b["Dense_0"]["bias"][wide]}
            b["Dense_1"] = {"kernel": b["Dense_1"]["kernel"][wide][:, trunk], "bias": b["Dense_1"]["bias"][trunk]}
            if "LayerNorm_0" in b:
                b["LayerNorm_0"] = {k: v[trunk] for k, v in b["LayerNorm_0"].items()}
    out[name] = pp
    return out


def float32_chain_deviation(c: Case, params, xT, cond, noise, ref, orders: int = 6):
    """What float32 arithmetic itself can promise on this chain: the largest deviation from the float64 oracle over
    `orders` float32 evaluations of the same network that differ only in summation order (hidden units renumbered).
    Returns (max rel err of x0 / xt_hist, rms of the same, max rel err of eps_hist).  An ill-conditioned schedule
    (cosine: 1/sqrt(alpha_hat_T) = 119) turns one evaluation's 5e-7 rounding noise into a heavy-tailed 1e-5..1e-4 on the
    worst element, so ONE float32 run is a poor estimate of that bar."""
    ref0, refx, refe = ref
    es, rs, ee = 0.0, 0.0, 0.0
    for k in range(orders):
        p = params if k == 0 else permuted_hidden_units(c, params, 7000 + k)
        o = oracle_run(c, p, xT, cond, noise, dtype=np.float32)
        es = max(es, rel_err(o[0], ref0), rel_err(o[1], refx))
        rs = max(rs, rms_err(o[0], ref0), rms_err(o[1], refx))
        ee = max(ee, rel_err(o[2], refe))
    return es, rs, ee


def product_model(c: Case, params, device="cuda"):
    """The flowrl_hip model object for a case (torch tensors on `device`)."""
    import flowrl_hip as fh
    from flowrl_hip.flow import DDPM, ContinuousDDPM, ContinuousNormalizingFlow
    te = None
    if c.temb == "positional":
        te = fh.PositionalEmbedding(output_dim=64)
    elif c.temb == "learnable_fourier":
        te = fh.LearnableFourierEmbedding(output_dim=64)
    if c.backbone == "residual":
        mlp = fh.ResidualMLP(hidden_dims=list(c.hidden), output_dim=c.A, activation=c.act, layer_norm=c.ln)
    elif c.backbone == "simba":
        mlp = fh.Simba(hidden_dims=list(c.hidden), output_dim=c.A, activation=c.act)
    else:
        mlp = fh.MLP(hidden_dims=list(c.hidden), output_dim=c.A, activation=c.act, layer_norm=c.ln, ln_after_act=c.ln_post)
    ce = fh.MLP(hidden_dims=[128, 128], activation=c.act) if c.cond_embed else None
    tparams = fh.tree_from_numpy(params, device)
    inputs = (np.ones((1, c.A)), np.zeros((1, 1)), np.ones((1, c.S)) if c.S > 0 else None)
    if c.kind == "ddpm":
        net = fh.DDPMBackbone(noise_predictor=mlp, time_embedding=te, cond_embedding=ce)
        return DDPM.create(net, 0, inputs, x_dim=c.A, steps=c.T, noise_schedule=c.schedule,
                           approx_postvar=c.approx_postvar, clip_sampler=c.clip, x_min=c.x_min, x_max=c.x_max,
                           params=tparams, device=device)
    if c.kind == "cddpm":
        net = fh.DDPMBackbone(noise_predictor=mlp, time_embedding=te, cond_embedding=ce)
        return ContinuousDDPM.create(net, 0, inputs, x_dim=c.A, steps=c.T, noise_schedule=c.schedule,
                                     clip_sampler=c.clip, x_min=c.x_min, x_max=c.x_max, solver=c.solver,
                                     params=tparams, device=device)
    net = fh.FlowBackbone(vel_predictor=mlp, time_embedding=te, cond_embedding=ce)
    return ContinuousNormalizingFlow.create(net, 0, inputs, x_dim=c.A, steps=c.T, clip_sampler=c.clip,
                                            x_min=c.x_min, x_max=c.x_max, params=tparams, device=device)


def product_run(c: Case, model, xT, cond, noise, device="cuda", history=True, **extra):
    import torch
    x = torch.from_numpy(xT).to(device)
    cd = None if cond is None else torch.from_numpy(cond).to(device)
    nz = torch.from_numpy(noise).to(device)
    kw = dict(history=history, particles=c.particles if c.particles > 1 else None,
              compute_dtype="bf16" if c.bf16 else "f32", tile_policy=c.tile, **extra)
    if c.kind == "ddpm":
        _, x0, (xh, eh) = model.sample(None, x, cd, False, c.solver, noise=nz, **kw)
    elif c.kind == "cddpm":
        _, x0, (xh, eh) = model.sample(None, x, cd, False, c.solver, noise=nz, **kw)
    else:
        _, x0, (xh, eh) = model.sample(None, x, cd, **kw)
    torch.cuda.synchronize()
    f = lambda t: None if t is None else t.cpu().numpy()
    return f(x0), f(xh), f(eh)


def rel_err(got, ref):
    """max |got - ref| / max(|ref|, 1): the parity metric of SURVEY.md section 7 (hard parts)."""
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.max(np.abs(np.asarray(got, dtype=np.float64) - ref) / np.maximum(np.abs(ref), 1.0))) if ref.size else 0.0


def rms_err(got, ref):
    ref = np.asarray(ref, dtype=np.float64)
    if ref.size == 0:
        return 0.0
    d = (np.asarray(got, dtype=np.float64) - ref) / np.maximum(np.abs(ref), 1.0)
    return float(np.sqrt(np.mean(d * d)))


# ---- independent torch-autograd restatement of the time-branch VJP (cross-check of csrc/time_bwd.hip) ----
def _mish64(x):
    import torch
    return x * torch.tanh(torch.nn.functional.softplus(x))


def torch_time_branch_vjp(network, params, tvals, grad_tb, x_dim):
    """Gradients of everything that feeds the hoisted bias table tb[j] = bias_0 + tfeat(t_j) @ kernel_0[A:A+D']
    (ddpm.py:84-91: time_embedding -> MLP_0 (mish, mish) -> concat slice of Dense_0), given dL/dtb from the kernel.
    Batch-invariant, O(T D^2): a handful of (T, D) float64 matrix ops on the device, differentiated with autograd.
    Returns a partial grads tree (float32): MLP_0, time_embedding, and the predictor's Dense_0 bias / time rows."""
    import torch
    from flowrl_hip.module import LearnableFourierEmbedding
    te = network.time_embedding
    pname = network.predictor_name
    f64 = torch.float64
    g = grad_tb.to(f64)
    W0 = params[pname]["Dense_0"]["kernel"]
    t = tvals.to(f64).reshape(-1, 1)
    out: Dict = {pname: {"Dense_0": {"bias": g.sum(0).to(torch.float32)}}}
    if te is None:
        out[pname]["Dense_0"]["time_rows"] = (t.T @ g).to(torch.float32)
        return out
    D = te.output_dim
    leaves = {("MLP_0", f"Dense_{i}", n): params["MLP_0"][f"Dense_{i}"][n].detach().to(f64).requires_grad_(True)
              for i in range(2) for n in ("kernel", "bias")}
    if isinstance(te, LearnableFourierEmbedding):
        wk = params["time_embedding"]["kernel"].detach().to(f64).requires_grad_(True)
        leaves[("time_embedding", "kernel")] = wk
        f = 2.0 * torch.pi * t @ wk.T                                     # time_embedding.py:23-27
    else:
        half = D // 2
        fr = torch.exp(torch.arange(half, dtype=f64, device=t.device) * -(torch.log(torch.tensor(10000.0, dtype=f64)) / (half - 1)))
        f = t * fr                                                        # time_embedding.py:11-16
    temb = torch.cat([torch.cos(f), torch.sin(f)], dim=-1)
    h = _mish64(temb @ leaves[("MLP_0", "Dense_0", "kernel")] + leaves[("MLP_0", "Dense_0", "bias")])
    tfeat = _mish64(h @ leaves[("MLP_0", "Dense_1", "kernel")] + leaves[("MLP_0", "Dense_1", "bias")])
    wt = W0[x_dim:x_dim + D].detach().to(f64).requires_grad_(True)
    keys = list(leaves)
    grads = torch.autograd.grad(tfeat @ wt, [leaves[k] for k in keys] + [wt], grad_outputs=g)
    for k, gv in zip(keys, grads[:-1]):
        node = out
        for name in k[:-1]:
            node = node.setdefault(name, {})
        node[k[-1]] = gv.to(torch.float32)
    out[pname]["Dense_0"]["time_rows"] = grads[-1].to(torch.float32)
    return out


