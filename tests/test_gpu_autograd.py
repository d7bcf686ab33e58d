"""flowrl_hip.autograd — the torch.autograd bridge that stands where ``jax.grad`` stands in the reference
(module/model.py:78-86 around the losses of dql.py:93-123, dacer.py:141-162, ddpm.py:280-304, cnf.py:145-170).

The bridge adds no arithmetic: its forward IS the library's training forward and its backward IS the library's VJP.
So the tests check (1) that gradients taken through ``loss.backward()`` are bit-identical to the explicit
``sample_vjp`` / ``eval_vjp`` calls with the same cotangent, and (2) against the float64 oracle's manual backprop with
the bars of tests/test_gpu_vjp.py (max 1e-5, rms 2e-6 of the leaf's largest entry; single evaluation 2e-5).
PARITY UNPINNED by the reference (it has no tests for this path)."""
import numpy as np
import pytest

import flowrl_oracle as orc
import helpers as H

pytestmark = pytest.mark.gpu


def _flatten(tree, prefix=""):
    out = {}
    for k, v in tree.items():
        if isinstance(v, dict):
            out.update(_flatten(v, prefix + k + "/"))
        else:
            out[prefix + k] = v
    return out


SAMPLE_CASES = [
    H.Case("ag-dql", "ddpm", A=3, S=11, T=5, B=256, seed=31),
    H.Case("ag-dacer", "cddpm", A=17, S=376, T=20, B=3 * 16, particles=16, temb="learnable_fourier", schedule="cosine",
           cond_embed=True, seed=32),
    H.Case("ag-cnf", "cnf", A=8, S=20, T=10, B=64, temb=None, clip=False, seed=33),
]


@pytest.mark.parametrize("case", SAMPLE_CASES, ids=lambda c: c.name)
def test_grad_through_sample(case):
    """An actor loss L(x0) = sum(tanh(x0 Wq)) + mean(x0^2) (a stand-in critic in plain torch ops) differentiated through
    fa.sample: (a) bit-identical to sample_vjp fed with torch's own dL/dx0, (b) against the float64 oracle's VJP."""
    import torch
    import flowrl_hip.autograd as fa
    from test_gpu_vjp import oracle_vjp
    params, xT, cond, noise = H.make_inputs(case)
    model = H.product_model(case, params)
    x = torch.from_numpy(xT).cuda()
    c = None if cond is None else torch.from_numpy(cond).cuda()
    parts = case.particles if case.particles > 1 else None
    Wq = torch.from_numpy(np.random.default_rng(case.seed).standard_normal((case.A, 4)).astype(np.float32)).cuda()
    kw = dict(particles=parts)
    if case.kind != "cnf":
        kw["solver"] = case.solver
        if case.solver == "ddpm":
            kw["noise"] = torch.from_numpy(noise).cuda()

    def critic(x0):
        return torch.tanh(x0 @ Wq).sum() + (x0 * x0).mean()

    def loss_fn(p):
        _, x0, hist = fa.sample(model, None, x, c, params=p, **kw)
        return critic(x0), {"x0": x0, "hist": hist}

    (loss, aux), grads = fa.value_and_grad(loss_fn, model.params, has_aux=True)
    torch.cuda.synchronize()
    assert aux["hist"][0].shape == (case.T, case.B, case.A) and not aux["hist"][0].requires_grad

    # (a) the explicit path: training forward, torch's dL/dx0, the library's VJP
    _, x0e, he = model.sample(None, x, c, True, params=model.params, keep_for_vjp=True, **kw)
    x0l = x0e.detach().requires_grad_(True)
    g = torch.autograd.grad(critic(x0l), x0l)[0]
    bkw = {k: v for k, v in kw.items() if k in ("solver", "particles")}
    g_exp, _ = model.sample_vjp(None, g, c, params=model.params, ctx=he[2], **bkw)
    torch.cuda.synchronize()
    assert torch.equal(aux["x0"].detach(), x0e)
    fe, fb = _flatten(g_exp), _flatten(grads)
    assert set(fe) <= set(fb)
    for k in fe:
        assert torch.equal(fe[k], fb[k]), f"{case.name}: {k} differs between loss.backward() and sample_vjp"
    for k in set(fb) - set(fe):     # leaves the chain does not reach get zeros (jax.grad's convention)
        assert float(fb[k].abs().max()) == 0.0, k

    # (b) the float64 oracle with the same cotangent
    _, gref, _ = oracle_vjp(case, params, xT, cond, noise, g.cpu().numpy().astype(np.float64))
    ref = _flatten(gref)
    worst = 0.0
    for k in ref:
        r = np.asarray(ref[k], np.float64)
        d = fb[k].cpu().numpy().astype(np.float64) - r
        scale = max(np.abs(r).max(), 1e-30)
        m, rms = np.abs(d).max() / scale, np.sqrt(np.mean(d * d)) / scale
        worst = max(worst, m)
        assert m <= 1e-5 and rms <= 2e-6, f"{case.name}: {k} max {m:.2e} rms {rms:.2e}"
    print(f"{case.name}: loss {float(loss.detach()):.6f}; grads through autograd bit-equal to sample_vjp; worst leaf vs float64 oracle {worst:.2e}")


def test_grad_wrt_xT_and_keyed_rng():
    """dL/dxT flows to a grad-requiring xT; with a threefry key the call returns the advanced key (ddpm.py:188,204-206)
    and the in-kernel draws give the same gradients as the same draws passed as a tensor."""
    import torch
    import flowrl_hip.autograd as fa
    import threefry_oracle as tf
    case = H.Case("ag-key", "ddpm", A=3, S=11, T=5, B=200, seed=34)
    params, xT, cond, _ = H.make_inputs(case)
    model = H.product_model(case, params)
    c = torch.from_numpy(cond).cuda()
    key = tf.prng_key(11)
    keys, rng_ref = tf.sampler_key_chain("ddpm", "ddpm", case.T, key)
    z = np.stack([tf.normal(keys[j], (case.B, case.A)) for j in range(case.T)]).astype(np.float32)

    def run(rng, noise):
        x = torch.from_numpy(xT).cuda().requires_grad_(True)
        rng_out, x0, _ = fa.sample(model, rng, x, c, solver="ddpm", **({} if noise is None else {"noise": noise}))
        loss = (x0 ** 3).sum()
        gx, = torch.autograd.grad(loss, [x])
        return rng_out, x0.detach(), gx

    r1, a1, g1 = run(torch.from_numpy(key).cuda(), None)
    r2, a2, g2 = run(None, torch.from_numpy(z).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(r1.cpu().numpy(), rng_ref) and r2 is None
    assert float((a1 - a2).abs().max()) <= 2e-5 and float((g1 - g2).abs().max()) <= 2e-5 * float(g2.abs().max())
    assert float(g1.abs().max()) > 0.0
    with pytest.raises(NotImplementedError):
        fa.sample(model, None, torch.from_numpy(xT).cuda(), c.clone().requires_grad_(True), solver="ddim")


def test_grad_through_apply_is_the_loss_grads():
    """The eps-MSE loss of jit_update_ddpm (ddpm.py:280-304) written with fa.apply + torch ops against the fused
    ddpm_loss(want_grads=True) of the same draws, and against the oracle's backward."""
    import torch
    import flowrl_hip.autograd as fa
    import flowrl_oracle_vjp as ov
    import threefry_oracle as tf
    from flowrl_hip.flow.ddpm import ddpm_loss
    case = H.Case("ag-mse", "ddpm", A=3, S=11, T=5, B=256, seed=35)
    params, x0, cond, _ = H.make_inputs(case)
    model = H.product_model(case, params)
    c = torch.from_numpy(cond).cuda()
    kt = torch.from_numpy(tf.prng_key(3)).cuda()
    _, loss_f, aux = ddpm_loss(kt, model, torch.from_numpy(x0).cuda(), c, want_grads=True)

    def loss_fn(p):
        _, xt, t, eps = model.add_noise(kt, torch.from_numpy(x0).cuda())
        pred = fa.apply(model, xt, t, c, params=p)
        return ((pred - eps) ** 2).mean()

    loss, grads = fa.value_and_grad(loss_fn, model.params)
    torch.cuda.synchronize()
    assert abs(float(loss.detach()) - float(loss_f)) <= 1e-6 * max(1.0, abs(float(loss_f)))
    fa_, ff = _flatten(grads), _flatten(aux["grads"])
    assert set(fa_) == set(ff)
    p64 = orc.cast_tree(params, np.float64)
    pred_ref, cache = ov.backbone_fwd(case.spec, p64, aux["xt"].cpu().numpy().astype(np.float64), aux["t"].cpu().numpy().astype(np.float64),
                                      cond.astype(np.float64))
    _, g_ref = orc.mse(pred_ref, aux["eps"].cpu().numpy().astype(np.float64))
    gref: dict = {}
    ov.backbone_bwd(case.spec, p64, cache, g_ref, gref)
    ref = _flatten(gref)
    for k in ff:
        scale = float(ff[k].abs().max())
        assert float((fa_[k] - ff[k]).abs().max()) <= 2e-6 * max(scale, 1e-30), k      # torch's mean vs the fused MSE kernel: rounding only
        r = np.asarray(ref[k], np.float64)
        assert np.abs(fa_[k].cpu().numpy() - r).max() <= 2e-5 * max(np.abs(r).max(), 1e-30), k


def test_apply_gradient_trains():
    """Model.apply_gradient (module/model.py:78-86) end to end: Adam steps on the eps-MSE loss of jit_update_ddpm
    (ddpm.py:280-304) through fa.apply.  Step 1 is checked leaf by leaf against the Adam rule applied to the fused
    ddpm_loss(want_grads=True) gradients (at k = 1 the update is lr g / (|g| + eps)); the dropout key is split once per
    call (model.py:80); over the steps the loss on a FIXED draw goes down and the old parameters stay untouched."""
    import torch
    import flowrl_hip.autograd as fa
    import flowrl_hip.model as fm
    import threefry_oracle as tf
    from flowrl_hip.flow.ddpm import ddpm_loss
    case = H.Case("ag-train", "ddpm", A=3, S=11, T=5, B=512, seed=36)
    params, x0, cond, _ = H.make_inputs(case)
    model = H.product_model(case, params)
    c = torch.from_numpy(cond).cuda()
    x0t = torch.from_numpy(x0).cuda()
    kt = torch.from_numpy(tf.prng_key(5)).cuda()
    dkey = tf.prng_key(9)
    state = fm.TrainState.create(model.params, fm.adam(1e-3), dropout_rng=torch.from_numpy(dkey).cuda())
    before = {"/".join(p): v.clone() for p, v in fa.tree_leaves(model.params)}

    def loss_fn(p, dropout_rng):
        assert dropout_rng is not None and tuple(dropout_rng.shape) == (2,)
        _, xt, t, eps = model.add_noise(kt, x0t)
        l = ((fa.apply(model, xt, t, c, params=p) - eps) ** 2).mean()
        return l, {"loss": l.detach()}

    _, l0, aux = ddpm_loss(kt, model, x0t, c, want_grads=True)
    m1, state, info = fm.apply_gradient(model, state, loss_fn)
    torch.cuda.synchronize()
    assert abs(float(info["loss"]) - float(l0)) <= 1e-6 * max(1.0, abs(float(l0)))
    assert np.array_equal(state.dropout_rng.cpu().numpy(), tf.split(dkey, 2)[1])
    g = {"/".join(p): v for p, v in fa.tree_leaves(aux["grads"])}
    for path, v in fa.tree_leaves(m1.params):
        k = "/".join(path)
        want = before[k] - 1e-3 * g[k] / (g[k].abs() + 1e-8)
        big = g[k].abs() > 1e-4 * g[k].abs().max()        # where |g| is rounding noise, g / |g| is too
        assert float((v - want)[big].abs().max()) <= 2e-6, k
        assert torch.equal(dict(fa.tree_leaves(model.params))[path], before[k]), k      # functional update
    losses = [float(info["loss"])]
    m = m1
    for _ in range(12):
        m, state, info = fm.apply_gradient(m, state, loss_fn)
        losses.append(float(info["loss"]))
    torch.cuda.synchronize()
    print("ag-train: eps-MSE over 13 Adam steps:", " ".join(f"{l:.4f}" for l in losses))
    assert state.step == 13 and losses[-1] < 0.8 * losses[0] and np.isfinite(losses).all()


def test_no_grad_calls_take_the_plain_paths():
    """Under torch.no_grad (a target network, an evaluation) fa.sample / fa.apply run the plain sampler / evaluation: same
    return structure, no grad_fn, no stash."""
    import torch
    import flowrl_hip.autograd as fa
    case = H.Case("ag-nograd", "ddpm", A=3, S=11, T=5, B=96, seed=37)
    params, xT, cond, noise = H.make_inputs(case)
    model = H.product_model(case, params)
    x, c, nz = torch.from_numpy(xT).cuda(), torch.from_numpy(cond).cuda(), torch.from_numpy(noise).cuda()
    with torch.no_grad():
        rng, x0, (xh, eh) = fa.sample(model, None, x, c, solver="ddpm", noise=nz)
        out = fa.apply(model, x, torch.full((case.B, 1), 3, dtype=torch.int32, device="cuda"), c)
    _, x0_ref, (xh_ref, eh_ref) = model.sample(None, x, c, False, "ddpm", noise=nz, history=True)
    torch.cuda.synchronize()
    assert rng is None and x0.grad_fn is None and out.grad_fn is None
    assert torch.equal(x0, x0_ref) and torch.equal(xh, xh_ref) and torch.equal(eh, eh_ref)
    assert out.shape == (case.B, case.A) and torch.isfinite(out).all()


def test_second_backward_and_inplace_change_detection():
    """a second backward through the same graph (retain_graph=True) must run from the kept stash and give the
    same bits; an in-place change of a parameter leaf between forward and backward must be caught by the version counter."""
    import torch
    import flowrl_hip.autograd as fa
    case = H.Case("ag-twice", "ddpm", A=3, S=11, T=5, B=128, seed=35)
    params, xT, cond, noise = H.make_inputs(case)
    model = H.product_model(case, params)
    x, c, nz = (torch.from_numpy(v).cuda() for v in (xT, cond, noise))
    leaves = [leaf.detach().requires_grad_(True) for _, leaf in fa.tree_leaves(model.params)]
    p = fa.tree_build([path for path, _ in fa.tree_leaves(model.params)], leaves)
    _, x0, _ = fa.sample(model, None, x, c, params=p, solver="ddpm", noise=nz)
    loss = (x0 ** 2).sum()
    g1 = torch.autograd.grad(loss, leaves, retain_graph=True, allow_unused=True)
    g2 = torch.autograd.grad(loss, leaves, retain_graph=True, allow_unused=True)
    torch.cuda.synchronize()
    for a, b in zip(g1, g2):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)
    with torch.no_grad():
        leaves[0].add_(1.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        torch.autograd.grad(loss, leaves, allow_unused=True)
