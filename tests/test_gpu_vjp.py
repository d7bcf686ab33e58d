"""GPU parity of the chain VJP (frl_sampler_bwd through DDPM.sample_vjp) against the float64 manual-backprop
oracle (oracle/flowrl_oracle_vjp.py, itself pinned by finite differences and torch autograd in
tests/test_oracle_vjp.py).  PARITY UNPINNED by the reference (no tests / fixtures for this path).

Loss = sum(g * x0) with a random cotangent g.  Tolerance, per parameter leaf and for dL/dxT:
    max |hip - oracle64| <= 1e-5 * max |oracle64 leaf|      and      rms(hip - oracle64) <= 2e-6 * max |oracle64 leaf|
The gradient is a sum over B*T chain-steps of float32 products (per-workgroup partial tiles added in a FIXED slot order
by frl_bwd_reduce_kernel: no float atomics, two calls give the same bits) taken through a chain that amplifies rounding
by 1/sqrt(alpha_hat_T) = 12.5 per unit (see test_gpu_parity.py); measured on MI355X: max 3e-7..9e-7, rms 8e-8..2.2e-7
over the cases below.
"""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

TOL_MAX, TOL_RMS = 1e-5, 2e-6


def _flatten(tree, prefix=""):
    out = {}
    for k, v in tree.items():
        if isinstance(v, dict):
            out.update(_flatten(v, prefix + k + "/"))
        else:
            out[prefix + k] = v
    return out


def oracle_vjp(case, params, xT, cond, noise, g, xt_hist=None):
    """float64 oracle VJP of the case's sampler: (x0, grads tree, dL/dxT)."""
    import flowrl_oracle_vjp as ov
    import flowrl_oracle as orc
    cd = None if cond is None else np.repeat(cond, case.particles, axis=0)
    kw = dict(steps=case.T, clip_sampler=case.clip, x_min=case.x_min, x_max=case.x_max, xt_hist=xt_hist)
    if case.kind == "ddpm":
        tables = orc.ddpm_tables(case.T, case.schedule, approx_postvar=case.approx_postvar, dtype=np.float32)
        return ov.ddpm_sample_vjp(case.spec, params, tables, xT, cd, noise, g, solver=case.solver, **kw)
    if case.kind == "cddpm":
        tables = orc.continuous_tables(case.T, case.schedule, dtype=np.float32)
        return ov.continuous_ddpm_sample_vjp(case.spec, params, tables, xT, cd, noise, g, solver=case.solver, **kw)
    return ov.cnf_sample_vjp(case.spec, params, xT, cd, g, **kw)


def product_vjp(case, model, xT, cond, noise, g, compute_dtype="f32", stash=False, out=None):
    """Forward with history + VJP through the product; returns (grads tree, dL/dxT, xt_hist) as torch tensors.
    ``stash``: the forward is the TRAINING forward (frl_sampler_fwd_train, keep_for_vjp=True) and the backward reads its
    stash (frl_sampler_bwd with have_stash) instead of recomputing every step from xt_hist."""
    import torch
    x = torch.from_numpy(xT).cuda()
    c = None if cond is None else torch.from_numpy(cond).cuda()
    gt = torch.from_numpy(g).cuda()
    parts = case.particles if case.particles > 1 else None
    kw = dict(history=True, particles=parts, params=model.params, compute_dtype=compute_dtype, keep_for_vjp=stash)
    if case.kind == "cnf":
        _, x0, hist = model.sample(None, x, c, True, **kw)
        grads, gxT = model.sample_vjp(hist[0], gt, c, particles=parts, compute_dtype=compute_dtype, ctx=hist[2] if stash else None)
    else:
        nz = torch.from_numpy(noise).cuda() if case.solver == "ddpm" else None
        _, x0, hist = model.sample(None, x, c, True, case.solver, noise=nz, **kw)
        grads, gxT = model.sample_vjp(hist[0], gt, c, case.solver, particles=parts, compute_dtype=compute_dtype, ctx=hist[2] if stash else None)
    if out is not None:
        out.update(x0=x0, xt_hist=hist[0], eps_hist=hist[1])
    return grads, gxT, hist[0]


def _run(case, stash=False):
    import torch
    params, xT, cond, noise = H.make_inputs(case)
    rng = np.random.default_rng(77 + case.seed)
    g = rng.standard_normal((case.B, case.A)).astype(np.float32)
    x0_ref, gref, gx_ref = oracle_vjp(case, params, xT, cond, noise, g)
    model = H.product_model(case, params)
    fwd = {}
    grads, gxT, _ = product_vjp(case, model, xT, cond, noise, g, stash=stash, out=fwd)
    torch.cuda.synchronize()
    if stash:
        # the training forward IS the sampler: its x0 / history against the float64 oracle at the forward's own bar
        r0, rx, re = H.oracle_run(case, params, xT, cond, noise, dtype=np.float64)
        o32 = H.oracle_run(case, params, xT, cond, noise, dtype=np.float32)
        tol = max(1e-5, 4.0 * max(H.rel_err(o32[0], r0), H.rel_err(o32[1], rx)))
        assert H.rel_err(fwd["eps_hist"][0].cpu().numpy(), re[0]) <= 1e-5
        assert H.rel_err(fwd["x0"].cpu().numpy(), r0) <= tol and H.rel_err(fwd["xt_hist"].cpu().numpy(), rx) <= tol, case.name
    got = {k: v.cpu().numpy() for k, v in _flatten(grads).items()}
    ref = _flatten(gref)
    assert set(got) == set(ref), (sorted(got), sorted(ref))
    got["dL/dxT"], ref["dL/dxT"] = gxT.cpu().numpy(), gx_ref
    worst = []
    for k in sorted(ref):
        r = np.asarray(ref[k], np.float64)
        d = got[k].astype(np.float64) - r
        scale = max(np.abs(r).max(), 1e-30)
        worst.append((np.abs(d).max() / scale, np.sqrt(np.mean(d * d)) / scale, k))
        assert got[k].shape == r.shape, k
    emax = max(worst)
    erms = max((w[1], w[2]) for w in worst)
    print(f"{case.name}: worst leaf max-err {emax[0]:.2e} ({emax[2]}), worst rms {erms[0]:.2e} ({erms[1]})")
    H.record_margin("vjp_f32", case.name, "stash" if stash else "recompute", leaf_max=emax[0], leaf_rms=erms[0], bar_max=TOL_MAX, bar_rms=TOL_RMS,
                    leaf_max_frac_of_bar=emax[0] / TOL_MAX, leaf_rms_frac_of_bar=erms[0] / TOL_RMS)
    for m, r, k in worst:
        assert m <= TOL_MAX and r <= TOL_RMS, f"{case.name}: {k} max {m:.2e} rms {r:.2e}"


CASES = [
    H.Case("cfg5_dql_b256", "ddpm", A=3, S=11, T=5, B=256),
    H.Case("ragged_b70", "ddpm", A=3, S=11, T=5, B=70, seed=1),
    H.Case("particles", "ddpm", A=3, S=11, T=5, B=8 * 50, particles=50, seed=2),
    H.Case("silu_noclip", "ddpm", A=6, S=17, T=4, B=128, act="silu", clip=False, seed=3),
    H.Case("relu_fourier", "ddpm", A=8, S=24, T=3, B=96, act="relu", temb="learnable_fourier", seed=4),
    H.Case("two_hidden", "ddpm", A=3, S=11, T=5, B=64, hidden=(256, 256), seed=5),
    H.Case("one_hidden_nocond", "ddpm", A=2, S=0, T=6, B=64, hidden=(256,), clip=False, seed=6),
    H.Case("raw_time", "ddpm", A=3, S=11, T=5, B=64, temb=None, seed=7),
    H.Case("multi_tile", "ddpm", A=3, S=11, T=5, B=20000, seed=8),
    # the other update rules the sweep knows (DACER differentiates ContinuousDDPM.sample, dacer.py:141-162)
    H.Case("ddpm_ddim", "ddpm", A=3, S=11, T=5, B=96, solver="ddim", seed=9),
    H.Case("cddpm_cosine", "cddpm", A=6, S=17, T=20, B=64, temb="learnable_fourier", schedule="cosine", seed=10),
    H.Case("cddpm_ddim_noclip", "cddpm", A=3, S=11, T=8, B=70, temb="learnable_fourier", schedule="cosine", solver="ddim",
           clip=False, seed=11),
    H.Case("cnf_raw_t", "cnf", A=8, S=20, T=10, B=64, temb=None, clip=False, seed=12),
    H.Case("cnf_clip_positional", "cnf", A=3, S=11, T=6, B=40, clip=True, act="silu", seed=13),
    # the other activations of flowrl/functional/activation.py (round 2: every forward activation has its derivative)
    H.Case("gelu_tanh", "ddpm", A=3, S=11, T=5, B=96, act="gelu_tanh", seed=14),
    H.Case("tanh_cnf", "cnf", A=6, S=17, T=6, B=64, act="tanh", clip=False, seed=15),
    H.Case("elu_cddpm", "cddpm", A=3, S=11, T=8, B=64, act="elu", temb="learnable_fourier", schedule="linear", seed=16),
]


# DACER / QVPO shapes (dacer.py:141-162 differentiates ContinuousDDPM.sample with its cond_embedding): the wide operand build,
# which exists for the training forward + stash backward; the recompute path must refuse them loudly
WIDE_CASES = [
    H.Case("dacer_humanoid", "cddpm", A=17, S=376, T=20, B=3 * 16, particles=16, temb="learnable_fourier", schedule="cosine",
           cond_embed=True, seed=20),
    H.Case("dacer_batch_p1", "cddpm", A=17, S=376, T=20, B=70, temb="learnable_fourier", schedule="cosine", cond_embed=True, seed=21),
    H.Case("embed_small_act", "cddpm", A=6, S=17, T=8, B=40, temb="learnable_fourier", schedule="cosine", cond_embed=True, act="silu", seed=22),
    H.Case("wide_cond_no_embed", "ddpm", A=17, S=128, T=5, B=64, seed=23),
    H.Case("embed_gelu", "cddpm", A=6, S=17, T=6, B=40, temb="learnable_fourier", schedule="linear", cond_embed=True, act="gelu_tanh", seed=24),
    H.Case("wide_x_only", "ddpm", A=12, S=11, T=5, B=50, act="relu", seed=24),
    H.Case("embed_cnf", "cnf", A=8, S=39, T=6, B=2 * 8, particles=8, clip=False, cond_embed=True, seed=25),
]


@pytest.mark.parametrize("case", WIDE_CASES, ids=lambda c: c.name)
def test_vjp_wide_shapes_from_the_stash(case):
    import flowrl_hip as fh
    _run(case, True)
    params, xT, cond, noise = H.make_inputs(case)
    g = np.zeros((case.B, case.A), np.float32)
    with pytest.raises(fh.FlowRLHipUnsupported):       # from xt_hist alone: no wide recompute build
        product_vjp(case, H.product_model(case, params), xT, cond, noise, g, stash=False)


@pytest.mark.parametrize("stash", [False, True], ids=["recompute", "stash"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_vjp_matches_oracle(case, stash):
    """Both backward paths: from xt_hist alone (the sweep recomputes every step) and from the training forward's stash."""
    _run(case, stash)


def test_vjp_unsupported_is_loud():
    import torch
    import flowrl_hip as fh
    case = H.Case("fql-512", "ddpm", A=3, S=11, T=5, B=32, hidden=(512, 512), act="gelu_tanh")      # only the 256-wide geometry has a VJP
    params, xT, cond, noise = H.make_inputs(case)
    model = H.product_model(case, params)
    xh = torch.zeros((5, 32, 3), device="cuda")
    with pytest.raises(fh.FlowRLHipUnsupported):
        model.sample_vjp(xh, torch.zeros((32, 3), device="cuda"), torch.from_numpy(cond).cuda())


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_vjp_is_bitwise_reproducible(dtype):
    """No float atomics anywhere in the VJP: per-workgroup partial tiles + a fixed-order second stage.  Two
    calls on the same inputs give identical bits for every leaf and for dL/dxT, on a ragged batch that spreads over
    many workgroups of the dW kernel and several steps per workgroup."""
    import torch
    case = H.Case("repro", "ddpm", A=3, S=11, T=5, B=32 * 700 + 13, seed=77, bf16=(dtype == "bf16"))
    params, xT, cond, noise = H.make_inputs(case)
    g = np.random.default_rng(5).standard_normal((case.B, case.A)).astype(np.float32)
    model = H.product_model(case, params)
    runs = []
    for _ in range(3):
        grads, gx, _ = product_vjp(case, model, xT, cond, noise, g, compute_dtype=dtype)
        torch.cuda.synchronize()
        runs.append(({k: v.cpu().numpy() for k, v in _flatten(grads).items()}, gx.cpu().numpy()))
    for other in runs[1:]:
        for k, v in runs[0][0].items():
            assert np.array_equal(v, other[0][k]), f"{k} differs between two identical calls"
        assert np.array_equal(runs[0][1], other[1])


def test_vjp_against_committed_golden():
    """The product VJP against tests/golden/golden_cfg5_vjp.npz (every gradient leaf + dL/dxT): no oracle run."""
    from pathlib import Path
    import torch
    import flowrl_oracle as orc
    z = np.load(Path(__file__).resolve().parent / "golden" / "golden_cfg5_vjp.npz")
    case = H.Case("golden_cfg5_vjp", "ddpm", A=3, S=11, T=5, B=40, seed=107)
    params = orc.unflatten_tree({k[2:]: z[k] for k in z.files if k.startswith("p:")})
    model = H.product_model(case, params)
    grads, gxT, _ = product_vjp(case, model, z["xT"], z["cond"], z["noise"], z["g"])
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in _flatten(grads).items()}
    want = {k[2:]: z[k] for k in z.files if k.startswith("g:")}
    assert set(got) == set(want)
    got["dL/dxT"], want["dL/dxT"] = gxT.cpu().numpy(), z["g_xT"]
    for k, r in want.items():
        d = got[k].astype(np.float64) - r
        scale = max(np.abs(r).max(), 1e-30)
        assert np.abs(d).max() / scale <= TOL_MAX and np.sqrt(np.mean(d * d)) / scale <= TOL_RMS, k


@pytest.mark.parametrize("big", [
    H.Case("lat-train-big", "ddpm", A=3, S=11, T=5, B=20000, seed=41),
    H.Case("lat-train-big-dacer", "cddpm", A=17, S=376, T=6, B=20000, temb="learnable_fourier", schedule="cosine", cond_embed=True, seed=42),
], ids=lambda c: c.name)
def test_small_batch_training_forward_is_the_latency_build(big):
    """frl_sampler_fwd_train runs batches of up to 16 384 chains as the latency build with the stash kept (four waves per
    32-chain tile, csrc/sampler_lat.hip STASH; capi.hip lat_train_ok) and larger ones as the one-wave-per-tile sweep.  Every
    small case above goes through the former against the float64 oracle; here the two meet: the first 300 chains of a
    20 000-chain call (throughput sweep) and the same 300 chains alone (latency build) must give the same x0, history and
    dL/dxT rows to float32 rounding — the stash the backward reads is laid out identically by both writers.  The DACER-shape
    case adds the wide operand build, the per-state hoist of the condition rows in the latency forward and the dL/dcfeat
    accumulation of the latency sweep."""
    import dataclasses
    import torch
    params, xT, cond, noise = H.make_inputs(big)
    g = np.random.default_rng(5).standard_normal((big.B, big.A)).astype(np.float32)
    model = H.product_model(big, params)
    fb, fs = {}, {}
    _, gx_big, _ = product_vjp(big, model, xT, cond, noise, g, stash=True, out=fb)
    n = 300
    small = dataclasses.replace(big, B=n)
    _, gx_small, _ = product_vjp(small, model, xT[:n], cond[:n], np.ascontiguousarray(noise[:, :n]), g[:n], stash=True, out=fs)
    torch.cuda.synchronize()
    e_x0 = H.rel_err(fs["x0"].cpu().numpy(), fb["x0"].cpu().numpy()[:n])
    e_h = max(H.rel_err(fs["xt_hist"].cpu().numpy(), fb["xt_hist"].cpu().numpy()[:, :n]), H.rel_err(fs["eps_hist"].cpu().numpy(), fb["eps_hist"].cpu().numpy()[:, :n]))
    gb = gx_big.cpu().numpy()[:n]
    e_gx = float(np.abs(gx_small.cpu().numpy() - gb).max() / np.abs(gb).max())
    print(f"{big.name}: latency vs throughput training forward: x0 {e_x0:.2e}, history {e_h:.2e}, dL/dxT rows {e_gx:.2e}")
    tol = 2e-5 if big.kind == "ddpm" else 1e-4      # the continuous chain divides by sqrt(alpha_hat) down to 0.08
    assert e_x0 <= tol and e_h <= tol and e_gx <= 1.5 * tol
