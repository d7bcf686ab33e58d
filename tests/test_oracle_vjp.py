"""CPU tests pinning the VJP oracle (oracle/flowrl_oracle_vjp.py): central finite differences in float64 and
torch autograd over an independent restatement of the whole chain."""
import math

import numpy as np
import pytest
import torch

import flowrl_oracle as orc
import flowrl_oracle_vjp as ov
import helpers as H


def _setup(act="mish", temb="positional", clip=True, B=6, T=4, hidden=(32, 32), A=3, S=5, seed=0):
    spec = orc.BackboneSpec(hidden_dims=list(hidden), x_dim=A, activation=act, time_embedding=temb, time_dim=16)
    params = orc.cast_tree(orc.make_params(spec, S, seed=seed), np.float64)
    rng = np.random.default_rng(seed + 1)
    xT, cond = rng.standard_normal((B, A)), rng.standard_normal((B, S))
    noise, g = rng.standard_normal((T, B, A)), rng.standard_normal((B, A))
    tabs = [t.astype(np.float64) for t in orc.ddpm_tables(T, "vp")]
    kw = dict(steps=T, clip_sampler=clip, x_min=-1.0, x_max=1.0)
    return spec, params, tabs, xT, cond, noise, g, kw


@pytest.mark.parametrize("act", ["mish", "silu", "tanh", "gelu_tanh", "elu"])
@pytest.mark.parametrize("temb", ["positional", "learnable_fourier"])
def test_vjp_matches_finite_differences(act, temb):
    spec, params, tabs, xT, cond, noise, g, kw = _setup(act=act, temb=temb, clip=False)
    x0, grads, gx = ov.ddpm_sample_vjp(spec, params, tabs, xT, cond, noise, g, **kw)
    flat, gflat = orc.flatten_tree(params), orc.flatten_tree(grads)
    assert set(flat) == set(gflat)

    def loss(p):
        return float(np.sum(g * orc.ddpm_sample(spec, p, tabs, xT, cond, noise, **kw)[0]))

    rng = np.random.default_rng(5)
    h = 1e-6
    for name, arr in flat.items():
        for _ in range(2):
            idx = tuple(rng.integers(0, s) for s in arr.shape)
            old = arr[idx]
            arr[idx] = old + h
            lp = loss(orc.unflatten_tree(flat))
            arr[idx] = old - h
            lm = loss(orc.unflatten_tree(flat))
            arr[idx] = old
            fd = (lp - lm) / (2 * h)
            assert abs(fd - gflat[name][idx]) <= 2e-6 * max(1.0, abs(fd)), (name, idx, fd, gflat[name][idx])
    # d loss / d xT
    xp, xm = xT.copy(), xT.copy()
    xp[1, 2] += h
    xm[1, 2] -= h
    fd = (float(np.sum(g * orc.ddpm_sample(spec, params, tabs, xp, cond, noise, **kw)[0]))
          - float(np.sum(g * orc.ddpm_sample(spec, params, tabs, xm, cond, noise, **kw)[0]))) / (2 * h)
    assert abs(fd - gx[1, 2]) <= 2e-6 * max(1.0, abs(fd))


def test_vjp_matches_torch_autograd_with_clip():
    spec, params, tabs, xT, cond, noise, g, kw = _setup(act="mish", temb="positional", clip=True, hidden=(64, 64, 64), B=16, T=5)
    x0, grads, gx = ov.ddpm_sample_vjp(spec, params, tabs, xT, cond, noise, g, **kw)
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in orc.flatten_tree(params).items()}
    xt = torch.tensor(xT, requires_grad=True)
    x = xt
    betas, alphas, ah, pv = [torch.tensor(t) for t in tabs]
    half = 8
    f = torch.exp(torch.arange(half, dtype=torch.float64) * -(math.log(10000.0) / (half - 1)))
    for j, t in enumerate(range(kw["steps"], 0, -1)):
        te = torch.cat([torch.cos(t * f), torch.sin(t * f)]).expand(x.shape[0], -1)
        for i in range(2):
            te = torch.nn.functional.mish(te @ tp[f"MLP_0/Dense_{i}/kernel"] + tp[f"MLP_0/Dense_{i}/bias"])
        hcur = torch.cat([x, te, torch.tensor(cond)], -1)
        for i in range(3):
            hcur = torch.nn.functional.mish(hcur @ tp[f"noise_predictor/Dense_{i}/kernel"] + tp[f"noise_predictor/Dense_{i}/bias"])
        eps = hcur @ tp["noise_predictor/Dense_3/kernel"] + tp["noise_predictor/Dense_3/bias"]
        x0h = torch.clamp((x - torch.sqrt(1 - ah[t]) * eps) / torch.sqrt(ah[t]), -1.0, 1.0)
        x = (torch.sqrt(ah[t - 1]) * (1 - alphas[t]) * x0h + torch.sqrt(alphas[t]) * (1 - ah[t - 1]) * x) / (1 - ah[t])
        x = x + float(t > 1) * torch.sqrt(pv[t]) * torch.tensor(noise[j])
    (torch.tensor(g) * x).sum().backward()
    np.testing.assert_allclose(x0, x.detach().numpy(), rtol=1e-10, atol=1e-12)
    for name, gv in orc.flatten_tree(grads).items():
        np.testing.assert_allclose(gv, tp[name].grad.numpy(), rtol=1e-8, atol=1e-10, err_msg=name)
    np.testing.assert_allclose(gx, xt.grad.numpy(), rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("kind,solver,clip", [("ddpm", "ddim", False), ("cddpm", "ddpm", False), ("cddpm", "ddim", False),
                                               ("cnf", "euler", False), ("cddpm", "ddpm", True)])
def test_vjp_other_samplers_match_finite_differences(kind, solver, clip):
    """ddim, ContinuousDDPM (what DACER differentiates, dacer.py:141-162) and CNF Euler: the same finite-difference pin."""
    temb = None if kind == "cnf" else "learnable_fourier"
    spec, params, tabs, xT, cond, noise, g, kw = _setup(act="mish", temb=temb, clip=clip)
    T = kw["steps"]
    if kind == "ddpm":
        fwd = lambda p, x: orc.ddpm_sample(spec, p, tabs, x, cond, noise, solver=solver, **kw)[0]
        x0, grads, gx = ov.ddpm_sample_vjp(spec, params, tabs, xT, cond, noise, g, solver=solver, **kw)
    elif kind == "cddpm":
        ctabs = [t.astype(np.float64) for t in orc.continuous_tables(T, "cosine")]
        fwd = lambda p, x: orc.continuous_ddpm_sample(spec, p, ctabs, x, cond, noise, solver=solver, **kw)[0]
        x0, grads, gx = ov.continuous_ddpm_sample_vjp(spec, params, ctabs, xT, cond, noise, g, solver=solver, **kw)
    else:
        spec = orc.BackboneSpec(hidden_dims=[32, 32], x_dim=3, activation="mish", time_embedding=None, time_dim=16,
                                predictor_name="vel_predictor")
        params = orc.cast_tree(orc.make_params(spec, 5, seed=0), np.float64)
        fwd = lambda p, x: orc.cnf_sample(spec, p, x, cond, **kw)[0]
        x0, grads, gx = ov.cnf_sample_vjp(spec, params, xT, cond, g, **kw)
    np.testing.assert_allclose(x0, fwd(params, xT), rtol=1e-6, atol=1e-7)     # (cnf: float32 time grid vs float64)
    flat, gflat = orc.flatten_tree(params), orc.flatten_tree(grads)
    assert set(flat) == set(gflat)
    rng = np.random.default_rng(7)
    h = 1e-6
    tol = 2e-5 if clip else 2e-6          # a finite-difference stencil may straddle a clip boundary
    bad = 0
    for name, arr in flat.items():
        for _ in range(2):
            idx = tuple(rng.integers(0, s) for s in arr.shape)
            old = arr[idx]
            arr[idx] = old + h
            lp = float(np.sum(g * fwd(orc.unflatten_tree(flat), xT)))
            arr[idx] = old - h
            lm = float(np.sum(g * fwd(orc.unflatten_tree(flat), xT)))
            arr[idx] = old
            fd = (lp - lm) / (2 * h)
            ok = abs(fd - gflat[name][idx]) <= tol * max(1.0, abs(fd))
            bad += not ok
            assert ok or clip, (name, idx, fd, gflat[name][idx])
    assert bad <= 1
    xp, xm = xT.copy(), xT.copy()
    xp[1, 2] += h
    xm[1, 2] -= h
    fd = (float(np.sum(g * fwd(params, xp))) - float(np.sum(g * fwd(params, xm)))) / (2 * h)
    assert abs(fd - gx[1, 2]) <= tol * max(1.0, abs(fd))


def test_vjp_oracle_reproduces_golden():
    """tests/golden/golden_cfg5_vjp.npz (made by tests/golden/make_golden.py): pins the manual-backprop oracle."""
    from pathlib import Path
    import flowrl_oracle as orc
    import flowrl_oracle_vjp as ov
    import helpers as H
    z = np.load(Path(__file__).resolve().parent / "golden" / "golden_cfg5_vjp.npz")
    case = H.Case("golden_cfg5_vjp", "ddpm", A=3, S=11, T=5, B=40, seed=107)
    params = orc.unflatten_tree({k[2:]: z[k] for k in z.files if k.startswith("p:")})
    tables = orc.ddpm_tables(case.T, case.schedule, dtype=np.float32)
    x0, grads, gx = ov.ddpm_sample_vjp(case.spec, params, tables, z["xT"], z["cond"], z["noise"], z["g"], solver="ddpm",
                                       steps=case.T, clip_sampler=True, x_min=-1.0, x_max=1.0)
    np.testing.assert_allclose(x0, z["x0"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(gx, z["g_xT"], rtol=1e-9, atol=1e-12)
    flat = orc.flatten_tree(grads)
    want = {k[2:]: z[k] for k in z.files if k.startswith("g:")}
    assert set(flat) == set(want)
    for k in want:
        np.testing.assert_allclose(flat[k], want[k], rtol=1e-9, atol=1e-12, err_msg=k)


@pytest.mark.parametrize("kw", [
    dict(hidden_dims=[24, 24], layer_norm=True),
    dict(hidden_dims=[24, 24, 24], layer_norm=True, ln_after_act=True, activation="gelu_tanh", time_embedding=None),
    dict(hidden_dims=[16, 16], backbone="residual", layer_norm=True),
    dict(hidden_dims=[16], backbone="residual", activation="silu"),
    dict(hidden_dims=[16, 16], backbone="simba", activation="relu", time_embedding="learnable_fourier"),
], ids=["ln-pre", "ln-post-fql", "residual-ln", "residual", "simba"])
def test_predictor_backward_matches_finite_differences(kw):
    """The LayerNorm / ResidualMLP / Simba backward of oracle/flowrl_oracle_vjp.py (the checker of the layer-by-layer
    single-evaluation VJP) against central finite differences of the float64 forward oracle, every leaf and dL/dx."""
    spec = orc.BackboneSpec(x_dim=3, time_dim=8, **{"activation": "mish", **kw})
    rng = np.random.default_rng(5)
    params = orc.cast_tree(orc.make_params(spec, 4, seed=3), np.float64)
    B = 6
    x, cond = rng.standard_normal((B, 3)), rng.standard_normal((B, 4))
    t = rng.uniform(0.05, 0.95, (B, 1)) if spec.time_embedding != "positional" else rng.integers(1, 6, (B, 1)).astype(np.float64)
    g = rng.standard_normal((B, 3))
    out, cache = ov.backbone_fwd(spec, params, x, t, cond)
    assert np.allclose(out, orc.backbone_apply(spec, params, x, t, cond), rtol=1e-12, atol=1e-13)
    grads: dict = {}
    gx = ov.backbone_bwd(spec, params, cache, g, grads)
    loss = lambda p, xx: float(np.sum(g * orc.backbone_apply(spec, p, xx, t, cond)))
    flat = orc.flatten_tree(params)
    gflat = orc.flatten_tree(grads)
    assert set(gflat) == set(flat), (sorted(gflat), sorted(flat))
    h = 1e-6
    for name, leaf in flat.items():
        idxs = [tuple(rng.integers(0, s) for s in leaf.shape) for _ in range(4)]
        for idx in idxs:
            pp = {k: v.copy() for k, v in flat.items()}
            pm = {k: v.copy() for k, v in flat.items()}
            pp[name][idx] += h
            pm[name][idx] -= h
            fd = (loss(orc.unflatten_tree(pp), x) - loss(orc.unflatten_tree(pm), x)) / (2 * h)
            assert abs(fd - gflat[name][idx]) <= 1e-6 * max(1.0, abs(fd)), (name, idx, fd, gflat[name][idx])
    for idx in [(0, 0), (3, 2), (5, 1)]:
        xp, xm = x.copy(), x.copy()
        xp[idx] += h
        xm[idx] -= h
        fd = (loss(params, xp) - loss(params, xm)) / (2 * h)
        assert abs(fd - gx[idx]) <= 1e-6 * max(1.0, abs(fd)), (idx, fd, gx[idx])


def test_oracle_reproduces_fql512_eval_golden():
    """tests/golden/golden_eval_vjp_fql512x4.npz (made by make_golden.py) is what the float64 oracle gives today."""
    import sys
    from pathlib import Path
    gdir = Path(__file__).resolve().parent / "golden"
    sys.path.insert(0, str(gdir))
    import make_golden as mg
    z = np.load(gdir / "golden_eval_vjp_fql512x4.npz")
    c = mg.EVAL_VJP_512
    params, _, _, _ = H.make_inputs(c)
    assert mg.params_digest(params) == str(z["params_sha256"])
    p64 = orc.cast_tree(params, np.float64)
    out, cache = ov.backbone_fwd(c.spec, p64, z["x"].astype(np.float64), z["t"].astype(np.float64), z["cond"].astype(np.float64))
    grads: dict = {}
    gx = ov.backbone_bwd(c.spec, p64, cache, z["g"].astype(np.float64), grads)
    assert np.allclose(out, z["out"], rtol=1e-12, atol=1e-13) and np.allclose(gx, z["g_x"], rtol=1e-12, atol=1e-13)
    for k, v in orc.flatten_tree(grads).items():
        v = np.asarray(v).reshape(-1)
        assert np.allclose(v[mg.leaf_sample_indices(k, v.size)], z["gs:" + k], rtol=1e-11, atol=1e-13), k
