"""GPU parity of the residual backbones (ResidualMLP flowrl/module/mlp.py:37-98, Simba
flowrl/module/simba.py:10-63) as the noise / velocity predictor of every sampler, against the float64 oracle
(oracle/flowrl_oracle.py residual_mlp_apply / simba_apply).  PARITY UNPINNED by the reference (no tests / fixtures).

Bars as tests/test_gpu_parity.py: first evaluation |hip - oracle64| <= 1e-5 max(|oracle64|, 1); chain state
max(1e-5, 4 x what float32 itself deviates by over 6 summation orders of the same network)."""
import numpy as np
import pytest

import flowrl_oracle as orc
import helpers as H

pytestmark = pytest.mark.gpu

CASES = [
    # DAC / BDPO with diffusion.resnet (dac.py:200-207, bdpo.py:325-332): ResidualMLP, mish, layer_norm=True, positional time
    H.Case("dac-resnet-2x256", "ddpm", A=3, S=11, T=5, B=300, hidden=(256, 256), backbone="residual", ln=True),
    H.Case("bdpo-resnet-3x256", "ddpm", A=6, S=17, T=5, B=70, hidden=(256, 256, 256), backbone="residual", ln=True, seed=1),
    H.Case("resnet-no-ln-ddim", "ddpm", A=3, S=11, T=5, B=64, hidden=(256,), backbone="residual", solver="ddim", act="relu", seed=2),
    H.Case("resnet-nocond-raw-t", "cnf", A=2, S=0, T=8, B=33, hidden=(256, 256), backbone="residual", temb=None, clip=False, act="silu", seed=3),
    # backbone_cls: simba (dacer.py:247-266 etc.): Simba behind the Fourier embedding and the cond_embedding, particles
    H.Case("simba-dacer-embed-particles", "cddpm", A=17, S=376, T=20, B=6 * 8, particles=8, hidden=(256, 256), backbone="simba",
           temb="learnable_fourier", schedule="cosine", cond_embed=True, act="relu", seed=4),
    H.Case("simba-cnf-genpo", "cnf", A=6, S=17, T=10, B=100, hidden=(256,), backbone="simba", clip=True, act="mish", seed=5),
    H.Case("simba-dpmpp2m", "cddpm", A=6, S=17, T=10, B=64, hidden=(256, 256), backbone="simba", temb="learnable_fourier",
           schedule="linear", solver="dpm++2m", act="gelu_tanh", seed=6),
    H.Case("resnet-multi-tile", "ddpm", A=3, S=11, T=5, B=20000, hidden=(256, 256), backbone="residual", ln=True, seed=7),
]


@pytest.mark.parametrize("build", ["throughput", "latency"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_residual_sampler(case, build):
    """Both builds of csrc/sampler_res.hip: one wave per 32-chain tile, and the latency build (four waves per tile, the
    expansion groups of a block split over them, partial sums added in wave order)."""
    import dataclasses
    import flowrl_hip as fh
    case = dataclasses.replace(case, tile=build)
    params, xT, cond, noise = H.make_inputs(case)
    ref = H.oracle_run(case, params, xT, cond, noise, dtype=np.float64)
    model = H.product_model(case, params)
    x0, xh, eh = H.product_run(case, model, xT, cond, noise)
    assert ("split 4" if build == "latency" else "split 1") in fh.last_kernels()
    e_first = H.rel_err(eh[0], ref[2][0])
    es, rs, ee = H.float32_chain_deviation(case, params, xT, cond, noise, ref)
    e0, ex = H.rel_err(x0, ref[0]), H.rel_err(xh, ref[1])
    print(f"{case.name}: first eval {e_first:.2e}; x0 {e0:.2e} xt_hist {ex:.2e} (float32 itself {es:.2e})")
    assert x0.shape == ref[0].shape and e_first <= 1e-5
    tol = max(1e-5, 4.0 * es)
    assert e0 <= tol and ex <= tol and H.rel_err(eh, ref[2]) <= max(1e-5, 4.0 * ee)
    H.record_margin("residual", case.name, build, first_eval=e_first, x0=e0, bar=tol)


def test_residual_auto_policy_picks_the_latency_build_for_acting_batches():
    """tile_policy auto: a Simba / ResidualMLP policy acting at batch 1 .. a few thousand chains runs four waves per tile;
    a throughput-sized batch keeps one wave per tile.  Two calls of the latency build are bit-identical (fixed-order sums)."""
    import torch
    import flowrl_hip as fh
    small = CASES[0]
    params, xT, cond, noise = H.make_inputs(small)
    model = H.product_model(small, params)
    a = H.product_run(small, model, xT, cond, noise)
    assert "split 4" in fh.last_kernels()
    b = H.product_run(small, model, xT, cond, noise)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    big = H.Case("resnet-throughput-size", "ddpm", A=3, S=11, T=2, B=140000, hidden=(256,), backbone="residual", ln=True, seed=12)
    params, xT, cond, noise = H.make_inputs(big)
    H.product_run(big, H.product_model(big, params), xT, cond, noise, history=False)
    assert "split 1" in fh.last_kernels()


def test_residual_in_kernel_rng_and_determinism():
    """The keyed form (in-kernel threefry draws) equals the noise-tensor form on the same draws; two calls are bit-identical."""
    import torch
    import threefry_oracle as tf
    case = CASES[0]
    params, xT, cond, _ = H.make_inputs(case)
    model = H.product_model(case, params)
    key = tf.prng_key(77)
    x = torch.from_numpy(xT).cuda()
    c = torch.from_numpy(cond).cuda()
    r1, a1, _ = model.sample(torch.from_numpy(key).cuda(), x, c)
    r2, a2, _ = model.sample(torch.from_numpy(key).cuda(), x, c)
    torch.cuda.synchronize()
    assert torch.equal(a1, a2) and torch.equal(r1, r2)
    # the same draws as an explicit tensor: the scan's split sequence (ddpm.py:205)
    nz, rng = tf.sampler_noise("ddpm", "ddpm", case.T, key, (case.B, case.A))
    nz = nz.astype(np.float32)
    ref = H.oracle_run(case, params, xT, cond, nz, dtype=np.float64)
    assert np.array_equal(r1.cpu().numpy(), rng)
    assert H.rel_err(a1.cpu().numpy(), ref[0]) <= 2e-5


def test_residual_net_eval_per_sample_time():
    """frl_net_eval on a residual predictor (the evaluation of add_noise-style losses and onestep_sample)."""
    import torch
    case = H.Case("resnet-eval", "ddpm", A=3, S=11, T=5, B=200, hidden=(256, 256), backbone="residual", ln=True, seed=9)
    params, x, cond, _ = H.make_inputs(case)
    t = np.random.default_rng(3).integers(1, case.T + 1, size=(case.B, 1)).astype(np.int32)
    ref = orc.backbone_apply(case.spec, orc.cast_tree(params, np.float64), x.astype(np.float64), t.astype(np.float64), cond.astype(np.float64))
    model = H.product_model(case, params)
    got = model(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda(), torch.from_numpy(cond).cuda())
    torch.cuda.synchronize()
    assert H.rel_err(got.cpu().numpy(), ref) <= 1e-5


def test_residual_refusals():
    """Loud refusals: bf16, 16-chain latency tiles, the chain VJP, unequal block widths."""
    import torch
    from flowrl_hip import _lib
    case = CASES[0]
    params, xT, cond, noise = H.make_inputs(case)
    model = H.product_model(case, params)
    x, c, nz = torch.from_numpy(xT).cuda(), torch.from_numpy(cond).cuda(), torch.from_numpy(noise).cuda()
    with pytest.raises(_lib.FlowRLHipUnsupported):
        model.sample(None, x, c, noise=nz, compute_dtype="bf16")
    with pytest.raises(_lib.FlowRLHipUnsupported):
        model.sample(None, x, c, noise=nz, tile_policy="latency16")
    with pytest.raises(_lib.FlowRLHipUnsupported):
        model.sample(None, x, c, True, noise=nz, history=True, keep_for_vjp=True, params=model.params)


def test_residual_against_committed_golden():
    """ResidualMLP (DAC resnet) and Simba against tests/golden/golden_res_dac_resnet.npz / golden_simba_cnf.npz: no oracle run
    (parameters regenerated from the stored seed and checked against the stored SHA-256)."""
    from pathlib import Path
    from test_oracle import GOLDEN_RES, _golden_params
    for case in GOLDEN_RES:
        z = np.load(Path(__file__).resolve().parent / "golden" / f"{case.name}.npz")
        params = _golden_params(z, case)
        model = H.product_model(case, params)
        x0, xh, eh = H.product_run(case, model, z["xT"], z["cond"], z["noise"], history=True)
        msg = f"{case.name}: x0 {H.rel_err(x0, z['x0']):.2e} xt {H.rel_err(xh, z['xt_hist']):.2e} first eval {H.rel_err(eh[0], z['eps_hist'][0]):.2e}"
        print(msg)
        assert H.rel_err(eh[0], z["eps_hist"][0]) <= 1e-5, msg
        assert H.rel_err(x0, z["x0"]) <= 5e-5 and H.rel_err(xh, z["xt_hist"]) <= 5e-5, msg
