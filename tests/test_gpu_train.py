"""GPU parity of the training-side half of the flow models against the oracles:
per-sample-time network evaluation (frl_net_eval), jax.random uniform / randint, add_noise / linear_interpolation,
DDPM.onestep_sample, and the epsilon-MSE / flow-matching losses.  PARITY UNPINNED by the reference (no tests / fixtures).

Bars: network outputs |hip - oracle64| <= 1e-5 max(|oracle64|, 1) (the sampler's first-evaluation bar); integer draws
bit-exact; uniform bit-exact (the same float32 expression); elementwise float32 pieces <= 2e-6."""
import dataclasses

import numpy as np
import pytest

import flowrl_oracle as orc
import helpers as H
import threefry_oracle as tf

pytestmark = pytest.mark.gpu


def _t_for(case, rng, B):
    if case.kind == "ddpm":
        return rng.integers(1, case.T + 1, size=(B, 1)).astype(np.int32)
    if case.kind == "cddpm":
        return rng.uniform(1e-3, 0.9946, size=(B, 1)).astype(np.float32)
    return rng.uniform(0.0, 1.0, size=(B, 1)).astype(np.float32)


EVAL_CASES = [
    H.Case("eval-dql", "ddpm", A=3, S=11, T=5, B=300),
    H.Case("eval-toy-nocond", "ddpm", A=2, S=0, T=20, B=33, hidden=(256,), act="relu", seed=1),
    H.Case("eval-cddpm-fourier", "cddpm", A=6, S=17, T=20, B=70, temb="learnable_fourier", schedule="cosine", act="silu", seed=2),
    H.Case("eval-fql-512-raw-t", "cnf", A=8, S=39, T=10, B=100, hidden=(512, 512, 512, 512), act="gelu_tanh", temb=None, clip=False, seed=3),
    H.Case("eval-dacer-embed-particles", "cddpm", A=17, S=376, T=20, B=6 * 8, particles=8, temb="learnable_fourier", schedule="cosine",
           cond_embed=True, seed=4),
    H.Case("eval-layer-norm", "ddpm", A=3, S=11, T=5, B=64, ln=True, seed=5),
    H.Case("eval-large", "ddpm", A=3, S=11, T=5, B=20000, seed=6),
]


@pytest.mark.parametrize("case", EVAL_CASES, ids=lambda c: c.name)
def test_net_eval_per_sample_time(case):
    import torch
    params, xT, cond, _ = H.make_inputs(case)
    rng = np.random.default_rng(50 + case.seed)
    t = _t_for(case, rng, case.B)
    cd = None if cond is None else np.repeat(cond, case.particles, axis=0).astype(np.float64)
    ref = orc.backbone_apply(case.spec, orc.cast_tree(params, np.float64), xT.astype(np.float64), t.astype(np.float64), cd)
    r32 = orc.backbone_apply(case.spec, params, xT, t.astype(np.float32), None if cd is None else cd.astype(np.float32))
    model = H.product_model(case, params)
    x = torch.from_numpy(xT).cuda()
    c = None if cond is None else torch.from_numpy(cond).cuda()
    got = model(x, torch.from_numpy(t).cuda(), c, particles=case.particles if case.particles > 1 else None)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    e, e32 = H.rel_err(got, ref), H.rel_err(r32, ref)
    print(f"{case.name}: eval rel err {e:.2e} (numpy f32 {e32:.2e}), rms {H.rms_err(got, ref):.2e}")
    assert got.shape == ref.shape and e <= 1e-5 and H.rms_err(got, ref) <= max(1e-6, 2.0 * H.rms_err(r32, ref))


@pytest.mark.parametrize("impl", ["partitionable", "legacy"])
def test_uniform_randint_bit_exact(impl):
    import torch
    from flowrl_hip import random as fr
    key = tf.prng_key(20240)
    kt = torch.from_numpy(key).cuda()
    for shape, lo, hi in (((1000, 1), 1, 6), ((777,), 0, 1000), ((5, 3), -7, 9), ((64,), 4, 4), ((300000,), 1, 21)):
        got = fr.randint(kt, shape, lo, hi, impl).cpu().numpy()
        assert np.array_equal(got, tf.randint(key, shape, lo, hi, impl)), (shape, lo, hi)
    for shape, lo, hi in (((1000, 1), 1e-3, 0.9946), ((4097,), 0.0, 1.0), ((3, 5), -2.0, 3.0)):
        got = fr.uniform(kt, shape, lo, hi, impl).cpu().numpy()
        assert np.array_equal(got, tf.uniform(key, shape, lo, hi, impl)), (shape, lo, hi)


def test_add_noise_all_three_models():
    """rng in -> (rng, xt, t, eps / vel) out as ddpm.py:164-171, continuous_ddpm.py:107-115, cnf.py:85-90 compute them."""
    import torch
    import flowrl_hip as fh
    key = tf.prng_key(99)
    kt = torch.from_numpy(key).cuda()
    rng = np.random.default_rng(3)
    B, A = 500, 6
    x0 = rng.standard_normal((B, A)).astype(np.float32)
    x1 = rng.standard_normal((B, A)).astype(np.float32)
    # --- DDPM
    c = H.Case("an-ddpm", "ddpm", A=A, S=0, T=5, B=B)
    m = H.product_model(c, H.make_inputs(c)[0])
    r, xt, t, eps = m.add_noise(kt, torch.from_numpy(x0).cuda())
    k0, kt_, kn = tf.split(key, 3)
    t_ref, eps_ref = tf.randint(kt_, (B, 1), 1, 6), tf.normal(kn, (B, A))
    assert np.array_equal(r.cpu().numpy(), k0) and np.array_equal(t.cpu().numpy(), t_ref)
    assert np.abs(eps.cpu().numpy() - eps_ref).max() <= 4e-7 * max(1.0, np.abs(eps_ref).max())
    tables = orc.ddpm_tables(5, "vp", dtype=np.float32)
    want = orc.ddpm_add_noise([v.astype(np.float64) for v in tables], x0.astype(np.float64), t_ref, eps.cpu().numpy().astype(np.float64))
    assert H.rel_err(xt.cpu().numpy(), want) <= 2e-6
    # --- ContinuousDDPM (cosine and linear)
    for sched, tmax in (("cosine", 0.9946), ("linear", 1.0)):
        c = H.Case("an-cddpm", "cddpm", A=A, S=0, T=5, B=B, temb="learnable_fourier", schedule=sched)
        m = H.product_model(c, H.make_inputs(c)[0])
        r, xt, t, eps = m.add_noise(kt, torch.from_numpy(x0).cuda())
        t_ref = tf.uniform(kt_, (B, 1), 1e-3, tmax)
        assert np.array_equal(r.cpu().numpy(), k0) and np.array_equal(t.cpu().numpy(), t_ref)
        e_np = eps.cpu().numpy()
        want = orc.continuous_add_noise(sched, x0.astype(np.float64), t_ref.astype(np.float64), e_np.astype(np.float64))
        # sigma = sqrt(1 - alpha^2) cancels in float32 near t = epsilon (alpha -> 1): float32 itself is 1e-6..6e-6 off there,
        # the reference included -> the bar is 4 x the float32 oracle's own deviation
        w32 = orc.continuous_add_noise(sched, x0, t_ref, e_np)
        assert H.rel_err(xt.cpu().numpy(), want) <= max(2e-6, 4.0 * H.rel_err(w32, want)), sched
    # --- CNF
    c = H.Case("an-cnf", "cnf", A=A, S=0, T=5, B=B, temb=None, clip=False)
    m = H.product_model(c, H.make_inputs(c)[0])
    r, xt, t, vel = m.linear_interpolation(kt, torch.from_numpy(x0).cuda(), torch.from_numpy(x1).cuda())
    k0b, ktb = tf.split(key, 2)
    t_ref = tf.uniform(ktb, (B, 1), 0.0, 1.0)
    assert np.array_equal(r.cpu().numpy(), k0b) and np.array_equal(t.cpu().numpy(), t_ref)
    wx, wv = orc.linear_interpolation(x0.astype(np.float64), x1.astype(np.float64), t_ref.astype(np.float64))
    assert H.rel_err(xt.cpu().numpy(), wx) <= 2e-6 and H.rel_err(vel.cpu().numpy(), wv) <= 1e-6


@pytest.mark.parametrize("solver,sample_xt,N", [("ddpm", True, 4), ("ddim", True, 1), ("ddpm", False, 3)])
def test_onestep_sample(solver, sample_xt, N):
    """DDPM.onestep_sample (ddpm.py:223-276) end to end against the oracle chain on the oracle's draws for the same key."""
    import torch
    case = H.Case("onestep", "ddpm", A=3, S=11, T=5, B=200, seed=31)
    params, x0, cond, _ = H.make_inputs(case)
    key = tf.prng_key(4321)
    model = H.product_model(case, params)
    t_in = np.random.default_rng(8).integers(1, 6, size=(case.B, 1)).astype(np.int32)
    out = model.onestep_sample(torch.from_numpy(key).cuda(), torch.from_numpy(x0).cuda(), torch.from_numpy(cond).cuda(), False, N, solver,
                               sample_xt, None if sample_xt else torch.from_numpy(t_in).cuda())
    torch.cuda.synchronize()
    rng_o, rep, xt, rep_t1, t, eps_theta = [v.cpu().numpy() for v in out]
    # oracle: same split sequence (ddpm.py:236), same draws
    k_rng, k_sample, k_noise, _ = tf.split(key, 4)
    tables64 = [v.astype(np.float64) for v in orc.ddpm_tables(5, "vp", dtype=np.float32)]
    if sample_xt:
        _, kt_, kn = tf.split(k_sample, 3)
        t_ref = tf.randint(kt_, (case.B, 1), 1, 6)
        xt_ref = orc.ddpm_add_noise(tables64, x0.astype(np.float64), t_ref, tf.normal(kn, x0.shape).astype(np.float64))
    else:
        t_ref, xt_ref = t_in, x0.astype(np.float64)
    assert np.array_equal(rng_o, k_rng) and np.array_equal(t, t_ref)
    assert H.rel_err(xt, xt_ref) <= 2e-6
    eps_ref = orc.backbone_apply(case.spec, orc.cast_tree(params, np.float64), xt.astype(np.float64), t_ref.astype(np.float64), cond.astype(np.float64))
    assert H.rel_err(eps_theta, eps_ref) <= 1e-5
    noise = tf.normal(k_noise, (case.B, N, case.A)).astype(np.float64)
    rep_ref, rt1_ref = orc.ddpm_onestep_update(tables64, xt.astype(np.float64), eps_theta.astype(np.float64), t_ref, N, noise, solver=solver,
                                               clip_sampler=True, x_min=-1.0, x_max=1.0)
    assert rep.shape == (case.B, N, case.A) and np.array_equal(rep_t1, rt1_ref)
    assert H.rel_err(rep, rep_ref) <= 5e-6


def test_losses():
    """jit_update_ddpm / jit_update_continuous_ddpm / jit_update_flow_matching's loss value and its cotangent."""
    import torch
    from flowrl_hip.flow.cnf import flow_matching_loss
    from flowrl_hip.flow.continuous_ddpm import continuous_ddpm_loss
    from flowrl_hip.flow.ddpm import ddpm_loss
    key = tf.prng_key(5)
    kt = torch.from_numpy(key).cuda()
    rng = np.random.default_rng(12)
    for case, fn in ((H.Case("loss-ddpm", "ddpm", A=3, S=11, T=5, B=256, seed=41), ddpm_loss),
                     (H.Case("loss-cddpm", "cddpm", A=6, S=17, T=20, B=256, temb="learnable_fourier", schedule="cosine", seed=42), continuous_ddpm_loss),
                     (H.Case("loss-fm", "cnf", A=8, S=39, T=10, B=256, temb=None, clip=False, hidden=(512, 512), act="gelu_tanh", seed=43), flow_matching_loss)):
        params, xa, cond, _ = H.make_inputs(case)
        model = H.product_model(case, params)
        c = torch.from_numpy(cond).cuda()
        if case.kind == "cnf":
            xb = rng.standard_normal(xa.shape).astype(np.float32)
            _, loss, aux = fn(kt, model, torch.from_numpy(xa).cuda(), torch.from_numpy(xb).cuda(), c)
            pred, target, g = aux["vel_pred"], aux["vel"], aux["g_vel_pred"]
        else:
            _, loss, aux = fn(kt, model, torch.from_numpy(xa).cuda(), c)
            pred, target, g = aux["eps_pred"], aux["eps"], aux["g_eps_pred"]
        torch.cuda.synchronize()
        xt, t = aux["xt"].cpu().numpy(), aux["t"].cpu().numpy()
        pred_ref = orc.backbone_apply(case.spec, orc.cast_tree(params, np.float64), xt.astype(np.float64), t.astype(np.float64), cond.astype(np.float64))
        assert H.rel_err(pred.cpu().numpy(), pred_ref) <= 1e-5, case.name
        l_ref, g_ref = orc.mse(pred.cpu().numpy().astype(np.float64), target.cpu().numpy().astype(np.float64))
        assert abs(float(loss) - l_ref) <= 1e-6 * max(1.0, abs(l_ref)), case.name
        assert np.abs(g.cpu().numpy() - g_ref).max() <= 1e-6 * np.abs(g_ref).max(), case.name


# ---- the evaluation under jax.grad: frl_net_eval_train + frl_net_eval_bwd ------------------------------------------------
EVAL_VJP_CASES = [
    H.Case("evjp-dql", "ddpm", A=3, S=11, T=5, B=300, seed=60),
    H.Case("evjp-ragged-nocond", "ddpm", A=2, S=0, T=20, B=33, hidden=(256,), act="relu", seed=61),
    H.Case("evjp-cddpm-fourier", "cddpm", A=6, S=17, T=20, B=70, temb="learnable_fourier", schedule="cosine", act="silu", seed=62),
    H.Case("evjp-cnf-raw-t", "cnf", A=8, S=39, T=10, B=100, hidden=(256, 256), temb=None, clip=False, seed=63),
    H.Case("evjp-dacer-embed-particles", "cddpm", A=17, S=376, T=20, B=6 * 8, particles=8, temb="learnable_fourier", schedule="cosine",
           cond_embed=True, seed=64),
    H.Case("evjp-wide-cond", "ddpm", A=17, S=128, T=5, B=64, seed=65),
    H.Case("evjp-multi-tile", "ddpm", A=3, S=11, T=5, B=20000, seed=66),
    H.Case("evjp-gelu-embed", "cddpm", A=6, S=17, T=20, B=50, temb="learnable_fourier", schedule="linear", cond_embed=True, act="gelu_tanh", seed=67),
    H.Case("evjp-elu", "cnf", A=4, S=9, T=10, B=70, act="elu", clip=False, seed=68),
    # ---- the layer-by-layer path (csrc/net_vjp_layered.hip): what the fused 256-wide sweep does not carry ----
    # cfg3's FQL actor_bc under the flow-matching loss (cnf.py:145-170, fql.py:83-123,189-211): 48 -> 512 x 4 -> 8, gelu-tanh, raw t
    H.Case("evjp-fql-512x4", "cnf", A=8, S=39, T=10, B=256, hidden=(512, 512, 512, 512), act="gelu_tanh", temb=None, clip=False, seed=90),
    H.Case("evjp-fql-512x4-lnpost", "cnf", A=8, S=39, T=10, B=100, hidden=(512, 512, 512, 512), act="gelu_tanh", temb=None, clip=False,
           ln=True, ln_post=True, seed=91),
    H.Case("evjp-ln-pre-dql", "ddpm", A=3, S=11, T=5, B=300, ln=True, seed=92),
    H.Case("evjp-ln-pre-embed-particles", "cddpm", A=17, S=376, T=20, B=5 * 8, particles=8, temb="learnable_fourier", schedule="cosine",
           cond_embed=True, ln=True, act="silu", seed=93),
    # DAC with diffusion.resnet (dac.py:200-207): ResidualMLP, 2 blocks, mish + LayerNorm; and without LayerNorm
    H.Case("evjp-res-dac", "ddpm", A=3, S=11, T=5, B=256, hidden=(256, 256), backbone="residual", ln=True, seed=94),
    H.Case("evjp-res-noln-ragged", "ddpm", A=6, S=17, T=5, B=70, hidden=(256,), backbone="residual", act="silu", seed=95),
    # Simba behind DACER's embeddings (backbone_cls: simba), relu
    H.Case("evjp-simba-dacer", "cddpm", A=17, S=376, T=20, B=4 * 8, particles=8, hidden=(256, 256), backbone="simba", act="relu",
           temb="learnable_fourier", schedule="cosine", cond_embed=True, seed=96),
    H.Case("evjp-512-positional", "ddpm", A=3, S=11, T=5, B=5000, hidden=(512, 512), seed=97),      # split over the batch in the dW GEMMs
]


def _flatten(tree, prefix=""):
    out = {}
    for k, v in tree.items():
        if isinstance(v, dict):
            out.update(_flatten(v, prefix + k + "/"))
        else:
            out[prefix + k] = v
    return out


@pytest.mark.parametrize("case", EVAL_VJP_CASES, ids=lambda c: c.name)
def test_net_eval_vjp(case):
    """dL/d(every parameter leaf) and dL/dx of ONE per-sample-time evaluation against the float64 manual-backprop oracle
    (oracle/flowrl_oracle_vjp.py backbone_fwd / backbone_bwd).  Bars as the chain VJP's: per leaf max 1e-5, rms 2e-6 of the
    leaf's largest entry."""
    import torch
    import flowrl_oracle_vjp as ov
    params, x, cond, _ = H.make_inputs(case)
    rng = np.random.default_rng(150 + case.seed)
    t = _t_for(case, rng, case.B)
    g = rng.standard_normal((case.B, case.A)).astype(np.float32)
    p64 = orc.cast_tree(params, np.float64)
    cd = None if cond is None else np.repeat(cond, case.particles, axis=0).astype(np.float64)
    out_ref, cache = ov.backbone_fwd(case.spec, p64, x.astype(np.float64), t.astype(np.float64), cd)
    gref: dict = {}
    gx_ref = ov.backbone_bwd(case.spec, p64, cache, g.astype(np.float64), gref)
    model = H.product_model(case, params)
    parts = case.particles if case.particles > 1 else None
    c = None if cond is None else torch.from_numpy(cond).cuda()
    out, ctx = model(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda(), c, True, model.params, particles=parts, keep_for_vjp=True)
    grads, gx = model.eval_vjp(ctx, torch.from_numpy(g).cuda(), want_gx=True)
    grads2, _ = model.eval_vjp(ctx, torch.from_numpy(g).cuda())
    torch.cuda.synchronize()
    assert H.rel_err(out.cpu().numpy(), out_ref) <= 1e-5
    got = {k: v.cpu().numpy() for k, v in _flatten(grads).items()}
    ref = _flatten(gref)
    assert set(got) == set(ref), (sorted(got), sorted(ref))
    for k, v in _flatten(grads2).items():      # fixed-order reduce: a second backward from the same stash is bit-identical
        assert np.array_equal(v.cpu().numpy(), got[k]), k
    got["dL/dx"], ref["dL/dx"] = gx.cpu().numpy(), gx_ref
    worst = []
    for k in sorted(ref):
        r = np.asarray(ref[k], np.float64)
        assert got[k].shape == r.shape, k
        d = got[k].astype(np.float64) - r
        scale = max(np.abs(r).max(), 1e-30)
        worst.append((np.abs(d).max() / scale, np.sqrt(np.mean(d * d)) / scale, k))
    print(f"{case.name}: worst leaf max-err {max(worst)[0]:.2e} ({max(worst)[2]})")
    for m, r, k in worst:
        assert m <= 1e-5 and r <= 2e-6, f"{case.name}: {k} max {m:.2e} rms {r:.2e}"


def test_loss_grads():
    """jit_update_ddpm / continuous_ddpm / flow_matching with want_grads: aux["grads"] = d loss / d params, against the
    oracle's backward of the same evaluation with the oracle's own MSE cotangent."""
    import torch
    import flowrl_oracle_vjp as ov
    from flowrl_hip.flow.cnf import flow_matching_loss
    from flowrl_hip.flow.continuous_ddpm import continuous_ddpm_loss
    from flowrl_hip.flow.ddpm import ddpm_loss
    key = tf.prng_key(6)
    kt = torch.from_numpy(key).cuda()
    rng = np.random.default_rng(13)
    for case, fn in ((H.Case("lg-ddpm", "ddpm", A=3, S=11, T=5, B=256, seed=71), ddpm_loss),
                     (H.Case("lg-cddpm", "cddpm", A=6, S=17, T=20, B=256, temb="learnable_fourier", schedule="cosine", seed=72), continuous_ddpm_loss),
                     (H.Case("lg-fm", "cnf", A=8, S=39, T=10, B=256, temb=None, clip=False, hidden=(256, 256), seed=73), flow_matching_loss),
                     # FQL's own step: the flow-matching loss on the 512 x 4 gelu-tanh actor_bc (fql.py:83-123); DAC-resnet's eps-MSE step
                     (H.Case("lg-fm-fql512", "cnf", A=8, S=39, T=10, B=256, temb=None, clip=False, hidden=(512, 512, 512, 512), act="gelu_tanh", seed=74), flow_matching_loss),
                     (H.Case("lg-ddpm-resdac", "ddpm", A=3, S=11, T=5, B=256, hidden=(256, 256), backbone="residual", ln=True, seed=75), ddpm_loss)):
        params, xa, cond, _ = H.make_inputs(case)
        model = H.product_model(case, params)
        c = torch.from_numpy(cond).cuda()
        if case.kind == "cnf":
            xb = rng.standard_normal(xa.shape).astype(np.float32)
            _, loss, aux = fn(kt, model, torch.from_numpy(xa).cuda(), torch.from_numpy(xb).cuda(), c, want_grads=True)
            target = aux["vel"]
        else:
            _, loss, aux = fn(kt, model, torch.from_numpy(xa).cuda(), c, want_grads=True)
            target = aux["eps"]
        torch.cuda.synchronize()
        xt, t = aux["xt"].cpu().numpy().astype(np.float64), aux["t"].cpu().numpy().astype(np.float64)
        p64 = orc.cast_tree(params, np.float64)
        pred_ref, cache = ov.backbone_fwd(case.spec, p64, xt, t, cond.astype(np.float64))
        l_ref, g_ref = orc.mse(pred_ref, target.cpu().numpy().astype(np.float64))
        gref: dict = {}
        ov.backbone_bwd(case.spec, p64, cache, g_ref, gref)
        assert abs(float(loss) - l_ref) <= 1e-5 * max(1.0, abs(l_ref)), case.name
        got, ref = _flatten(aux["grads"]), _flatten(gref)
        assert set(got) == set(ref)
        for k in ref:
            r = np.asarray(ref[k], np.float64)
            d = got[k].cpu().numpy().astype(np.float64) - r
            assert np.abs(d).max() <= 2e-5 * max(np.abs(r).max(), 1e-30), (case.name, k, np.abs(d).max() / np.abs(r).max())


def test_net_eval_vjp_refusals():
    """Loud refusals: a missing gradient buffer is FRL_EINVAL; the bf16 variant has no single-evaluation VJP."""
    import ctypes as C
    import torch
    from flowrl_hip import _lib, _train
    case = H.Case("evjp-ln", "ddpm", A=3, S=11, T=5, B=64, ln=True, seed=80)
    params, x, cond, _ = H.make_inputs(case)
    model = H.product_model(case, params)
    t = _t_for(case, np.random.default_rng(1), case.B)
    out, ctx = model(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda(), torch.from_numpy(cond).cuda(), True, model.params, keep_for_vjp=True)
    bufs = _lib.FrlEvalBwdBufs()       # no gradient pointers at all
    g = torch.zeros((case.B, case.A), device="cuda")
    bufs.x, bufs.t, bufs.cond, bufs.g_out = ctx.x.data_ptr(), ctx.t.data_ptr(), ctx.cond2d.data_ptr(), g.data_ptr()
    lib = _lib.load()
    rc = lib.frl_net_eval_bwd(C.byref(ctx.desc), C.byref(ctx.fparams), C.byref(bufs), ctx.ws.data_ptr(), ctx.ws.numel(), 0)
    assert rc == _lib.FRL_EINVAL and b"grad_" in lib.frl_last_error()


def test_net_eval_vjp_against_committed_golden():
    """frl_net_eval_train / frl_net_eval_bwd against tests/golden/golden_eval_vjp.npz (every leaf + dL/dx): no oracle run."""
    import torch
    from pathlib import Path
    z = np.load(Path(__file__).resolve().parent / "golden" / "golden_eval_vjp.npz")
    case = H.Case("golden_eval_vjp", "cddpm", A=6, S=17, T=20, B=40, temb="learnable_fourier", schedule="cosine", act="silu", seed=111)
    params = orc.unflatten_tree({k[2:]: z[k] for k in z.files if k.startswith("p:")})
    model = H.product_model(case, params)
    out, ctx = model(torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["t"]).cuda(), torch.from_numpy(z["cond"]).cuda(), True, model.params,
                     keep_for_vjp=True)
    grads, gx = model.eval_vjp(ctx, torch.from_numpy(z["g"]).cuda(), want_gx=True)
    torch.cuda.synchronize()
    assert H.rel_err(out.cpu().numpy(), z["out"]) <= 1e-5
    got = {"g:" + k: v.cpu().numpy() for k, v in _flatten(grads).items()}
    assert set(got) == {k for k in z.files if k.startswith("g:")}
    got["g_x"] = gx.cpu().numpy()
    for k, v in got.items():
        r = z[k]
        assert np.abs(v - r).max() <= 1e-5 * max(np.abs(r).max(), 1e-30), k


def test_net_eval_vjp_fql_512x4_against_committed_golden():
    """The layer-by-layer VJP on cfg3's FQL actor_bc (48 -> 512 x 4 -> 8, gelu-tanh, raw t, FQL's post-activation LayerNorm)
    against tests/golden/golden_eval_vjp_fql512x4.npz — every gradient leaf at 4096 seeded entries plus its L2 norm, the
    parameters regenerated from their seed and checked by SHA-256: no oracle run."""
    import sys
    import torch
    from pathlib import Path
    gdir = Path(__file__).resolve().parent / "golden"
    sys.path.insert(0, str(gdir))
    import make_golden as mg
    z = np.load(gdir / "golden_eval_vjp_fql512x4.npz")
    case = mg.EVAL_VJP_512
    params, _, _, _ = H.make_inputs(case)
    assert mg.params_digest(params) == str(z["params_sha256"])
    model = H.product_model(case, params)
    out, ctx = model(torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["t"]).cuda(), torch.from_numpy(z["cond"]).cuda(), True, model.params,
                     keep_for_vjp=True)
    grads, gx = model.eval_vjp(ctx, torch.from_numpy(z["g"]).cuda(), want_gx=True)
    torch.cuda.synchronize()
    assert H.rel_err(out.cpu().numpy(), z["out"]) <= 1e-5
    assert np.abs(gx.cpu().numpy() - z["g_x"]).max() <= 1e-5 * np.abs(z["g_x"]).max()
    got = _flatten(grads)
    assert {"gs:" + k for k in got} == {k for k in z.files if k.startswith("gs:")}
    for k, v in got.items():
        v = v.cpu().numpy().astype(np.float64).reshape(-1)
        l2, mx = z["gn:" + k]
        assert abs(np.linalg.norm(v) - l2) <= 1e-5 * l2, k
        assert np.abs(v[mg.leaf_sample_indices(k, v.size)] - z["gs:" + k]).max() <= 1e-5 * mx, k
