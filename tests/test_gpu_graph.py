"""GPU: the entry points under HIP graph capture (torch.cuda.CUDAGraph).  Acting loops at small batch are launch-bound
(prepare + embedding + sampler + critic kernels of a few microseconds each); the library's calls are enqueue-only — no
allocation, no host synchronisation, no retained pointers (include/flowrl_hip.h) — so a whole acting step can be captured
once and replayed.  Checked: a replay on new inputs (copied into the captured tensors) returns bit for bit what the eager
call returns, for the plain sampler (caller workspace and the static-scratch entry point), the keyed call (the advanced rng
comes out of the graph) and the one-launch sample + select."""
import numpy as np
import pytest

import flowrl_oracle as orc
import helpers as H

pytestmark = pytest.mark.gpu


def _capture(fn, warmup=3):
    import torch
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warmup):          # first calls configure kernels (dynamic LDS sizes) outside the capture
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


@pytest.mark.parametrize("static_scratch", [False, True], ids=["workspace", "static"])
def test_sampler_replays_bit_identically(static_scratch):
    import torch
    c = H.Case("graph-ddpm", "ddpm", A=3, S=11, T=5, B=256, seed=41)
    params, xT, cond, noise = H.make_inputs(c)
    model = H.product_model(c, params)
    x, cd, nz = (torch.from_numpy(v).cuda() for v in (xT, cond, noise))
    g, out = _capture(lambda: model.sample(None, x, cd, False, "ddpm", noise=nz, static_scratch=static_scratch)[1])
    rng = np.random.default_rng(5)
    for _ in range(2):
        x2, c2, n2 = (rng.standard_normal(v.shape).astype(np.float32) for v in (xT, cond, noise))
        x.copy_(torch.from_numpy(x2)); cd.copy_(torch.from_numpy(c2)); nz.copy_(torch.from_numpy(n2))
        g.replay()
        torch.cuda.synchronize()
        got = out.cpu().numpy().copy()
        want = model.sample(None, x, cd, False, "ddpm", noise=nz, static_scratch=static_scratch)[1].cpu().numpy()
        np.testing.assert_array_equal(got, want)
        ref = H.oracle_run(c, params, x2, c2, n2, dtype=np.float64)[0]
        assert H.rel_err(got, ref) <= 1e-5


def test_keyed_call_replays_and_advances_the_rng():
    import torch
    import flowrl_hip.random as fr
    c = H.Case("graph-keyed", "cddpm", A=6, S=17, T=8, B=96, temb="learnable_fourier", schedule="cosine", seed=42)
    params, xT, cond, _ = H.make_inputs(c)
    model = H.product_model(c, params)
    x, cd = torch.from_numpy(xT).cuda(), torch.from_numpy(cond).cuda()
    key = fr.PRNGKey(7)

    def step():
        rng_out, x0, _ = model.sample(key, x, cd, False, "ddpm")
        return rng_out, x0

    g, (rng_out, x0) = _capture(step)
    for seed in (8, 9):
        key.copy_(fr.PRNGKey(seed))
        g.replay()
        torch.cuda.synchronize()
        got_rng, got_x0 = rng_out.cpu().numpy().copy(), x0.cpu().numpy().copy()
        want_rng, want_x0, _ = model.sample(fr.PRNGKey(seed), x, cd, False, "ddpm")
        np.testing.assert_array_equal(got_rng, want_rng.cpu().numpy())
        np.testing.assert_array_equal(got_x0, want_x0.cpu().numpy())


def test_fused_sample_and_select_replays():
    import torch
    import flowrl_hip as fh
    c = H.Case("graph-select", "ddpm", A=3, S=11, T=5, B=64 * 8, particles=8, seed=43)
    params, xT, cond, noise = H.make_inputs(c)
    model = H.product_model(c, params)
    cp = fh.tree_from_numpy(orc.make_critic_params(11, 3, (256, 256), 2, 44), "cuda")
    critic = fh.Ensemblize(base=fh.ScalarCritic(backbone=fh.MLP(hidden_dims=[256, 256], activation="mish")), ensemble_size=2)
    x = torch.from_numpy(xT.reshape(64, 8, 3)).cuda()
    cd, nz = torch.from_numpy(cond).cuda(), torch.from_numpy(noise).cuda()

    def step():
        return fh.sample_and_select(model, critic, cp, None, cd, False, 8, "ddpm", 0.0, xT=x, noise=nz, fused=True)[1]

    g, best = _capture(step)
    rng = np.random.default_rng(6)
    x.copy_(torch.from_numpy(rng.standard_normal((64, 8, 3)).astype(np.float32)))
    cd.copy_(torch.from_numpy(rng.standard_normal(cond.shape).astype(np.float32)))
    g.replay()
    torch.cuda.synchronize()
    got = best.cpu().numpy().copy()
    np.testing.assert_array_equal(got, step().cpu().numpy())
